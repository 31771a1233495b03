"""CPU sanitizer build of the library's host side (GPU AddressSanitizer is not available on the pool): every
source of music_amd/csrc compiled with AddressSanitizer + UndefinedBehaviorSanitizer on the HOST half (-fno-gpu-sanitize: the
device code is built as usual, so that the objects link and register like the product library), linked with
tests/host_sanitize.cpp, which walks the launch plans over a grid of shapes and the argument checks of the entry points.
No GPU is touched (the driver calls host-only paths)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_code_under_address_and_ub_sanitizers(tmp_path):
    flags = ["--offload-arch=gfx950", "-O1", "-g0", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
             "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-Wno-unused-function", "-Wno-unused-variable"]
    srcs = sorted(glob.glob(os.path.join(ROOT, "music_amd", "csrc", "*.hip")))
    procs = []
    for s in srcs:                                           # all at once (14 files; the device halves dominate: about a minute)
        o = str(tmp_path / (os.path.basename(s)[:-4] + ".o"))
        procs.append((s, o, subprocess.Popen([HIPCC] + flags + ["-c", s, "-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for s, o, p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, "%s:\n%s" % (s, out[-3000:])
        objs.append(o)
    exe = str(tmp_path / "host_sanitize")
    drv = str(tmp_path / "host_sanitize_main.o")
    r = subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=undefined", "-c", os.path.join(ROOT, "tests", "host_sanitize.cpp"), "-o", drv],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-gpu-sanitize", drv] + objs + ["-o", exe],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "host_sanitize ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    shutil.rmtree(tmp_path, ignore_errors=True)
