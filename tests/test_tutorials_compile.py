"""Boundary proof against the reference's own callers (SURVEY.md section 8b "What calls it"), build-container part:
the device code of the five tutorials BASELINE.json's configs name is compiled (-fsyntax-only, g++ -std=c++11, the flags of the
reference's AVX2 target) from the reference tree where it lies, against THIS repository's include/embree3/rtcore.h.
The tutorials include the API by a path relative to their own directory ("../../../include/embree3/rtcore.h",
tutorials/common/tutorial/tutorial_device.h:30); oracle/Makefile's `farm` target builds a directory tree of symlinks so that the
relative path resolves to our header (nothing is copied).  Skipped where /root/reference is absent (GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
FARM = os.path.join(ROOT, "oracle", "_ref", "farm")
TUTORIALS = ["triangle_geometry", "displacement_geometry", "viewer_stream", "pathtracer", "viewer"]

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tutorials")), reason="reference tree absent")


@pytest.fixture(scope="module")
def farm():
    subprocess.check_call(["make", "--no-print-directory", "-C", os.path.join(ROOT, "oracle"), "farm"], stdout=subprocess.DEVNULL)
    assert os.path.realpath(os.path.join(FARM, "include")) == os.path.join(ROOT, "include")
    return FARM


@pytest.mark.parametrize("name", TUTORIALS)
def test_tutorial_device_code_compiles_against_our_headers(farm, name):
    src = os.path.join(farm, "tutorials", name, name + "_device.cpp")
    base = ["g++", "-std=c++11", "-DTASKING_INTERNAL", "-mavx2", "-mfma", "-Wno-deprecated", "-w"]
    r = subprocess.run(base + ["-fsyntax-only", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    # the API header that was really used is ours
    deps = subprocess.run(base + ["-M", src], capture_output=True, text=True).stdout.replace("\\\n", " ").split()
    api = {os.path.realpath(d) for d in deps if d.endswith("embree3/rtcore.h")}
    assert api == {os.path.join(ROOT, "include", "embree3", "rtcore.h")}, api


def test_tutorial_binaries_link_against_our_library():
    """oracle/Makefile `tutorials`: triangle_geometry and displacement_geometry device code + the reference's own
    common/tasking and common/sys (the reference's library exports embree::TaskScheduler to its tutorials,
    kernels/export.linux.map:3; a tutorial built against this library links those two directories itself) + oracle/tut_harness.cpp
    link against embree-compressed_amd/lib/libembree3.so without unresolved symbols.  They run in tests/test_gpu_tutorials.py."""
    subprocess.check_call(["make", "--no-print-directory", "-C", os.path.join(ROOT, "oracle"), "tutorials"], stdout=subprocess.DEVNULL)
    for t in ("triangle_geometry", "displacement_geometry"):
        exe = os.path.join(ROOT, "oracle", "_ref", "tut_" + t)
        assert os.path.exists(exe)
        needed = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True).stdout
        assert "libembree3.so.3" in needed
