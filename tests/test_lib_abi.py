"""CPU test: the C-ABI shared library loads and exports every symbol include/avsr_hip.h declares,
and the ctypes table covers exactly that set (no compute call is made)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "avsr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(avsr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from mocov2_whisper_flamingo_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in avsr_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.load().avsr_version().startswith(b"avsr_hip")


def test_header_is_plain_c():
    import subprocess
    subprocess.run(["gcc", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "avsr_hip.h")], check=True)


def test_struct_layout_matches_header():
    """Every field of the three descriptor structs sits where the header puts it (offsets from gcc)."""
    from mocov2_whisper_flamingo_amd._lib import GemmDesc, XattnBlock, XattnGrads
    import subprocess, tempfile
    pairs = ((GemmDesc, "avsr_gemm_t"), (XattnBlock, "avsr_xattn_block_t"), (XattnGrads, "avsr_xattn_grads_t"))
    lines = []
    for cls, cname in pairs:
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
    src = "#include <stdio.h>\n#include <stddef.h>\n#include \"avsr_hip.h\"\nint main(void){\n" + "\n".join(lines) + "\nreturn 0; }\n"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        out = [int(x) for x in subprocess.run([os.path.join(d, "t")], check=True, capture_output=True, text=True).stdout.split()]
    got = []
    for cls, _ in pairs:
        got.append(ctypes.sizeof(cls))
        got.extend(getattr(cls, fname).offset for fname, _ in cls._fields_)
    assert out == got


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    from mocov2_whisper_flamingo_amd import ops, _lib
    with pytest.raises(_lib.AvsrHipError):
        ops.gemm_nt(torch.zeros(8, 8), torch.zeros(8, 8))
    # the product package never imports the oracle
    pkg = os.path.join(ROOT, "mocov2_whisper_flamingo_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, f)).read(), f
