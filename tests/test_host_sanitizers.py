"""The host code on the gen.phi path that indexes by pedigree data -- the native loader and the planner (levels, cuts, per-step arrays on
worker threads, hub walks) -- under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer ("use
-fsanitize=address for the host planner"; GPU sanitizers do not exist on this pool).  tests/host_sanitize.cpp links planner.cpp and
loader.cpp directly: no HIP, no oracle.  CPU only."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "genlib.jl_amd", "csrc")
DATA = os.path.join(ROOT, "genlib.jl_amd", "data")


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_planner_and_loader_under_sanitizers(san, tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    cmd = [gxx, "-std=c++17", "-O1", "-g", f"-fsanitize={san}", "-fno-omit-frame-pointer", "-pthread",
           os.path.join(ROOT, "tests", "host_sanitize.cpp"), os.path.join(CSRC, "planner.cpp"), os.path.join(CSRC, "loader.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and ("cannot find" in build.stderr or "unrecognized" in build.stderr):
        pytest.skip("this toolchain has no -fsanitize=" + san)
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", TSAN_OPTIONS="halt_on_error=1")
    env.pop("GENPHI_ENV_HOOKS", None)                               # (the planner's default thread counts)
    run = subprocess.run([exe, os.path.join(DATA, "genea140.csv"), os.path.join(DATA, "geneaJi.csv")], capture_output=True, text=True, env=env,
                         timeout=600)
    if run.returncode != 0 and not run.stdout and any(t in run.stderr for t in ("unexpected memory mapping", "runtime does not come first", "failed to intercept",
                                                                                     "ReserveShadowMemoryRange failed")):
        pytest.skip("the sanitizer runtime does not start in this environment: " + run.stderr[:200])
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-3000:])
    assert "host sanitizer run: ok" in run.stdout
    assert "runtime error" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr[-3000:]
