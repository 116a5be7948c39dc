"""The C99 example (BASELINE config 0, triangle_geometry plumbing) builds against include/embree3/rtcore.h on every
machine and, on the GPU box, runs through rtcIntersect1 / rtcOccluded1 / rtcIntersect1M with closed-form checks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "embree-compressed_amd", "lib")


def _build(tmp_path, name="triangle_geometry_min"):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=200112L", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".c"), "-L" + LIBDIR, "-lembree3", "-lm", "-lpthread",
           "-Wl,-rpath," + LIBDIR, "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_header_is_c99_and_example_links(tmp_path):
    _build(tmp_path)


def test_small_calls_example_links(tmp_path):
    _build(tmp_path, "small_calls_mt")


def test_cxx_header_compiles(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text('#include <embree3/rtcore.h>\n#include <embree3/rtcore_ray.h>\n#include <embree3/rtcore_amd.h>\n'
                   'int main() { RTCRayHitNt<4> p; (void)p; RTCSceneFlags f = RTC_SCENE_FLAG_ROBUST | RTC_SCENE_FLAG_COMPACT; '
                   'static_assert(sizeof(RTCRayHit) == 80 && sizeof(RTCRay) == 48 && sizeof(RTCBounds) == 32, "abi"); '
                   'static_assert(RTC_FORMAT_FLOAT3 == 0x9003 && RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR == 0x9234 && RTC_FORMAT_UINT3 == 0x5003, "abi"); '
                   'return f == 6 ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I" + os.path.join(ROOT, "include"), str(src), "-fsyntax-only"])


@pytest.mark.gpu
def test_triangle_geometry_example_runs(tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe, "gpu=0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout


@pytest.mark.gpu
def test_concurrent_rtcIntersect1_from_c_threads_is_combined(tmp_path):
    """Row f2: 48 pthreads x 400 single-ray calls; every answer checked in the program, and the calls must have been
    traced several per launch (C threads queue up behind the launch in flight; Python threads cannot show this)."""
    import re
    exe = _build(tmp_path, "small_calls_mt")
    out = subprocess.run([exe, "48", "400", "gpu=0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 mismatches" in out.stdout
    per_launch = float(re.search(r"\(([0-9.]+) calls per launch\)", out.stdout).group(1))
    print(out.stdout)
    assert per_launch > 4.0, out.stdout
