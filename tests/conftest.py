import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _source_hash():
    """sha256 over the library's sources, the same way embree-compressed_amd/Makefile records it in lib/BUILD_HASH."""
    import glob
    import hashlib
    pkg = os.path.join(ROOT, "embree-compressed_amd")
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.cpp")) + glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "embree3", "*.h")), key=lambda f: os.path.relpath(f, pkg))
    h = hashlib.sha256()
    for f in files:
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _ensure_built():
    lib = os.path.join(ROOT, "embree-compressed_amd", "lib", "libembree3.so")
    orc = os.path.join(ROOT, "oracle", "liboracle.so")
    stamp = os.path.join(ROOT, "embree-compressed_amd", "lib", "BUILD_HASH")
    # a prebuilt library travels to the GPU box with the snapshot; it is rebuilt there only when it does not match the sources
    stale = not os.path.exists(stamp) or open(stamp).read().strip() != _source_hash()
    if not os.path.exists(lib) or stale:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "embree-compressed_amd"), "-j8"], stdout=subprocess.DEVNULL)
    if not os.path.exists(orc):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


_ensure_built()


@pytest.fixture(scope="session")
def rtc():
    return importlib.import_module("embree-compressed_amd").rtc


@pytest.fixture(scope="session")
def po():
    import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def bomberman():
    d = np.load(os.path.join(ROOT, "assets", "bomberman.mesh.npz"))
    return d["verts"], d["face_sizes"], d["face_index"]


@pytest.fixture(scope="session")
def bomberman_tris(bomberman, rtc):
    v, fs, fi = bomberman
    return v, rtc.fan_triangulate(fs, fi)


@pytest.fixture(scope="session")
def gpu_device(rtc):
    dev = rtc.Device("")
    yield dev
    dev.release()
