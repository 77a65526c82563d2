"""CPU: libmt4hip.so loads and exports every symbol include/mt4hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "mt4hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mt4_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound():
    from computervision_codes_amd import _lib
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(_lib.lib, n), f"{n} declared in mt4hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_error_strings():
    from computervision_codes_amd import _lib
    assert _lib.lib.mt4_abi_version() == _lib.ABI_VERSION == 10
    assert _lib.lib.mt4_strerror(0) == b"ok"
    assert b"invalid" in _lib.lib.mt4_strerror(-1)


def test_argument_validation_without_gpu():
    """error paths return before any launch"""
    from computervision_codes_amd import _lib
    assert _lib.lib.mt4_conv_nhwc(None, None) == -1
    d = _lib.ConvDesc()
    assert _lib.lib.mt4_conv_nhwc(ctypes.byref(d), None) == -1
    assert _lib.lib.mt4_linear_f32(None, None, None, None, 1, 1, 1, None) == -1
    # packed K: fp32 512ch k3 = 3 taps x 128 chunks x 4; bf16 stem view = 32 chunks x 8
    assert _lib.lib.mt4_conv_packed_k(512, 1, 3, 0) == 1536
    assert _lib.lib.mt4_conv_packed_k(8, 7, 4, 1) == 256
    assert _lib.lib.mt4_conv_packed_k(48, 1, 1, 0) == 64   # 12 chunks -> 16 chunks x 4


def test_ops_refuse_cpu_tensors():
    import torch
    from computervision_codes_amd import _lib, ops
    with pytest.raises(_lib.Mt4Error):
        ops.maxpool3x3s2(torch.zeros(1, 4, 4, 4))


def test_struct_layout_matches_header():
    from computervision_codes_amd import _lib
    # 6 pointers + 24 int32 + fuse_cout + the three fuse pointers (8-byte aligned: 25 int32 pad to 104 bytes) + fuse_relu + residual_float
    # + the second K source: x2 and its four int32, + fuse_expand (padded to 8 bytes) + stat_sums
    assert ctypes.sizeof(_lib.ConvDesc) == 6 * 8 + 26 * 4 + 3 * 8 + 2 * 4 + 8 + 4 * 4 + 8 + 8
    assert _lib.ConvDesc.stat_sums.offset == ctypes.sizeof(_lib.ConvDesc) - 8
    assert _lib.ConvDesc.fuse_w.offset == 6 * 8 + 26 * 4 and _lib.ConvDesc.fuse_relu.offset == 6 * 8 + 26 * 4 + 24


def test_integration_doc_struct_matches_header_mirror():
    """the ctypes example a reference maintainer would copy from INTEGRATION.md lists exactly the fields of mt4_conv_desc"""
    import os
    import re
    from computervision_codes_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    block = doc[doc.index("class ConvDesc"):doc.index("lib.mt4_conv_nhwc.argtypes")]
    assert re.findall(r'"([A-Za-z_][A-Za-z_0-9]*)"', block) == [f[0] for f in _lib.ConvDesc._fields_]
    hdr = open(os.path.join(root, "include", "mt4hip.h")).read()
    struct = hdr[hdr.index("typedef struct mt4_conv_desc {") + len("typedef struct mt4_conv_desc {"):hdr.index("} mt4_conv_desc;")]
    struct = re.sub(r"/\*.*?\*/", "", struct, flags=re.S)
    fields = []
    for decl in struct.split(";"):
        m = re.match(r"\s*(?:const\s+)?(?:void|float|double|int32_t)\s*\*?\s*(.+)$", decl.strip().replace("\n", " "))
        if m:
            fields += [n.strip() for n in m.group(1).split(",")]
    assert fields == [f[0] for f in _lib.ConvDesc._fields_], fields


def test_loaded_library_is_built_from_the_sources_in_the_tree(tmp_path):
    """`mt4_source_digest()` (baked by csrc/Makefile) == `srcdigest.library_digest()` of the sources on disk; a library beside CHANGED sources is
    refused at import (MT4_ALLOW_STALE=1 loads it anyway)"""
    import shutil
    import subprocess
    import sys
    from computervision_codes_amd import _lib, srcdigest
    assert _lib.lib.mt4_source_digest().decode() == srcdigest.library_digest() == _lib.SOURCE_DIGEST and len(_lib.SOURCE_DIGEST) == 16
    assert {"igemm_conv.hip", "mt4_common.h", "../../include/mt4hip.h"} <= set(srcdigest.library_sources())
    pkg = tmp_path / "computervision_codes_amd"
    shutil.copytree(os.path.join(ROOT, "computervision_codes_amd"), pkg, ignore=shutil.ignore_patterns("*.o", "__pycache__"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    with open(pkg / "csrc" / "tcn_kernels.hip", "a") as f:
        f.write("\n// edited after the build\n")
    code = "import computervision_codes_amd._lib as l; print('loaded', l.SOURCE_DIGEST)"
    env = {k: v for k, v in os.environ.items() if k != "MT4_ALLOW_STALE"}
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "rebuild" in r.stderr and "MT4_ALLOW_STALE" in r.stderr, r.stderr[-500:]
    r = subprocess.run([sys.executable, "-c", code], cwd=tmp_path, env=dict(env, MT4_ALLOW_STALE="1"), capture_output=True, text=True)
    assert r.returncode == 0 and "loaded " + _lib.SOURCE_DIGEST in r.stdout, r.stderr[-500:]
