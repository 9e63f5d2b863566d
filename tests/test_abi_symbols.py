"""CPU: libmirt.so loads and exports every symbol include/mirt.h declares; with no GPU the
product refuses to run (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mirt.h")).read()
    return sorted(set(re.findall(r"MIRT_API\s+[\w\s\*]+?\b(mirt_\w+)\s*\(", text)))


def test_header_and_binding_agree(pkg):
    from raytracing_amd.pyhost import mirt
    assert declared_symbols() == sorted(mirt.SYMBOLS)


def test_library_exports_every_declared_symbol(pkg):
    from raytracing_amd.pyhost import mirt
    lib = mirt.lib()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.mirt_version()


def test_abi_version_matches_the_header(pkg):
    from raytracing_amd.pyhost import mirt
    text = open(os.path.join(ROOT, "include", "mirt.h")).read()
    assert mirt.lib().mirt_abi_version() == int(re.search(r"#define MIRT_ABI_VERSION (\d+)", text).group(1))


def test_no_cpu_fallback(pkg):
    from raytracing_amd.pyhost import mirt
    if mirt.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(mirt.MirtError) as e:
        mirt.Context(0)
    assert e.value.code == -7 and "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """Nothing under 2015-raytracing_amd/ may include, link or import oracle/."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "2015-raytracing_amd")):
        for f in files:
            if f.endswith((".so", ".node", ".o", ".pyc")):
                continue
            t = open(os.path.join(d, f), errors="ignore").read()
            if re.search(r"oracle/|liboracle|a10_pass|cl_numerics\.h|pt_oracle", t):
                bad.append(os.path.join(d, f))
    assert bad == []
