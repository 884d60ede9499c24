"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/iisan_hip.h declares,
the ctypes mirror covers exactly those symbols, and the ctypes struct layouts equal the C compiler's."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iisan_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(iisan_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from iisan_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/iisan_hip.h but not exported by libiisan_hip.so"
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    assert lib.iisan_arch() == b"gfx950" and b"iisan_hip" in lib.iisan_version()


def test_ctypes_struct_layout_matches_c(tmp_path):
    from iisan_amd import _lib
    prog = tmp_path / "sz.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "iisan_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                    'sizeof(iisan_layer_weights),sizeof(iisan_vit_weights),sizeof(iisan_bert_weights),sizeof(iisan_side_cfg),'
                    'sizeof(iisan_sasrec_cfg),offsetof(iisan_vit_weights,layer),offsetof(iisan_bert_weights,layer));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(_lib.LayerWeights), C.sizeof(_lib.VitWeights), C.sizeof(_lib.BertWeights), C.sizeof(_lib.SideCfg),
            C.sizeof(_lib.SasrecCfg), _lib.VitWeights.layer.offset, _lib.BertWeights.layer.offset]
    assert got == want, (got, want)


def test_ops_refuse_cpu_tensors_loudly():
    import torch
    from iisan_amd import _lib, ops
    with pytest.raises(_lib.IisanHipError):
        ops.LinearFn.apply(torch.zeros(4, 8), torch.zeros(3, 8), torch.zeros(3))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from iisan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IisanHipError):
        _lib.load()
