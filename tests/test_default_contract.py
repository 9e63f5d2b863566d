"""GPU: the SECOND numerics contract -- the reference as its own host builds it.

A10 code.js:599 calls program.build() without options: AMD's default 2.5-ulp division and 3-ulp sqrt.  The pin of every other test adds
-cl-fp32-correctly-rounded-divide-sqrt (the one contract a CPU checker can reproduce); libmirt_default.so (csrc/build.sh) is the same source built
for the default one -- every `/` and sqrt the sequence AMD's OpenCL compiler emits for the reference's text without the option.  Device against
device: the reference's code.cl compiled by AMD's OpenCL toolchain with its defaults (oracle/_ref/a10_gfx950_default.hsaco) and libmirt_default.so,
same seeds, every accumulator, seed and pixel.  The check runs in a process of its own: a process loads one libmirt."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

DEFAULT_HSACO = os.path.join(ROOT, "oracle", "_ref", "a10_gfx950_default.hsaco")
DEFAULT_LIB = os.path.join(ROOT, "2015-raytracing_amd", "libmirt_default.so")
needs = pytest.mark.skipif(not (os.path.exists(DEFAULT_HSACO) and os.path.exists(DEFAULT_LIB)), reason="needs the default-build code object and libmirt_default.so")


def run_check(*which):
    env = dict(os.environ, MIRT_CONTRACT="default")
    env.pop("MIRT_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "default_contract_check.py"), *which], env=env, capture_output=True, text=True, timeout=900)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.gpu
@needs
def test_every_scene_equals_the_default_build_of_the_reference():
    """all thirteen fixture scenes at 480x270 x 16 rays per pixel, depth 8, two progressive passes"""
    r, lines = run_check("scenes")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert len(lines) == 13 and all(l["ok"] for l in lines)


@pytest.mark.gpu
@needs
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "a07_gfx950_default.hsaco")), reason="needs the default builds of Assign01/04/07 (make -C oracle ref_gpu)")
def test_frame_kernels_equal_the_default_builds_of_the_reference():
    """BASELINE configs 1-3 (Assign01 / 04 / 07) and the molecule mode at full size, twelve frames: every pixel and every ray's maxt"""
    r, lines = run_check("frames")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert len(lines) == 12 and all(l["ok"] for l in lines)


@pytest.mark.gpu
@needs
def test_pixels_of_1024_rays_equal_the_default_build_of_the_reference():
    """config 5's ray count: the block-by-block in-pass resolve (four launches, no accumulator) under the default contract, depth 8, two scenes"""
    r, lines = run_check("big_pixels")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert len(lines) == 2 and all(l["ok"] for l in lines)


@pytest.mark.gpu
@needs
def test_kernel_by_kernel_path_equals_the_default_build_of_the_reference():
    """the fourteen kernels one by one (fusion off) in the default-contract library: every buffer after a pass -- every Ray, shadow Ray (the stored t of
    a blocked one included: it caught the compiler folding a single-use reciprocal into its product), vertex, accumulator, seed, pixel -- on four fixture
    scenes and 24 generated ones"""
    r, lines = run_check("granular")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert len(lines) == 28 and all(l["ok"] for l in lines)


@pytest.mark.gpu
@needs
def test_generated_scenes_equal_the_default_build_of_the_reference():
    """64 generated scenes (MIRT_SOAK=N: N) with grids of 1..7 cells per axis: the optimistic pair -- since round 4 this library has one: the optimistic
    kernel's structure with every quotient a plain division -- the exact kernel alone, and the in-pass resolve, against the reference's default build"""
    r, lines = run_check("random")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert lines and lines[-1]["ok"]


@pytest.mark.gpu
@needs
def test_headline_frame_equals_the_default_build_of_the_reference():
    """cornell.xml 1920x1080 x 256 rays per pixel, depth 8: 530 841 600 samples"""
    r, lines = run_check("headline")
    assert r.returncode == 0, (lines[-1:] or r.stderr[-2000:])
    assert lines and lines[-1]["ok"] and lines[-1]["samples"] == 530841600


def test_the_default_contract_library_exports_the_same_abi(pkg):
    """CPU: libmirt_default.so is a drop-in file -- every symbol of include/mirt.h, the same ABI version."""
    import ctypes as C
    from raytracing_amd.pyhost import mirt
    if not os.path.exists(DEFAULT_LIB):
        pytest.skip("libmirt_default.so not built")
    lib = C.CDLL(DEFAULT_LIB)
    for name in mirt.SYMBOLS:
        assert hasattr(lib, name), name
    lib.mirt_abi_version.restype = C.c_int
    assert lib.mirt_abi_version() == mirt.lib().mirt_abi_version()
