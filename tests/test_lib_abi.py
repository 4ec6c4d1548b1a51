"""The C-ABI library builds for gfx950, loads, and exports every symbol that
include/han_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

from han_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "han_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(han_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    names = _declared()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names


def test_library_loads_and_exports_every_symbol():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in _declared():
        assert hasattr(lib, n), n
    bound = _lib.load()
    assert bound.han_abi_version() == _lib.ABI_VERSION
    assert b"bad argument" in bound.han_error_string(-1)
    # size queries are host-only and safe without a GPU
    assert bound.han_project_bwd_workspace(1000, 256, 8, 8) >= 256 * 64 * 4
    assert bound.han_sem_attn_bwd_workspace(10, 2, 64, 128) > 0
    assert bound.han_classifier_workspace(10, 64, 3, 1) > 0
    assert bound.han_node_attn_bwd_workspace(10, 8, 8) > 0
    assert bound.han_score_param_bwd_workspace(10, 8, 8) > 0


def test_code_object_is_gfx950():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_row_split_struct_layout_matches_the_header(tmp_path):
    """han_row_split_t (ABI 6: + the degree bins) as plain C sees it == the ctypes mirror, field by field."""
    import subprocess
    fields = [f for f, _ in _lib.HanRowSplit._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "han_hip.h"\nint main(void) {\n'
                   '  printf("%zu\\n", sizeof(han_row_split_t));\n'
                   + "".join(f'  printf("%zu\\n", offsetof(han_row_split_t, {f}));\n' for f in fields)
                   + '  printf("%d %d\\n", HAN_ABI_VERSION, HAN_SHORT_DEG);\n  return 0;\n}\n')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_lib.HanRowSplit)
    for f, off in zip(fields, out[1:]):
        assert int(off) == getattr(_lib.HanRowSplit, f).offset, f
    assert int(out[-2]) == _lib.ABI_VERSION
    from han_amd import ops
    assert int(out[-1]) == ops.SHORT_DEG
