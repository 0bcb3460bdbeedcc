"""The C-ABI library loads and exports every symbol include/prach.h declares (no compute calls), and
refuses loudly to simulate without a device (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "prach.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(prach_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    syms = declared_symbols()
    assert len(syms) >= 16
    L = C.CDLL(pkg.LIB_PATH)
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(pkg.EXPORTS) == syms


def test_struct_sizes_match_header(pkg, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include "prach.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu\\n", sizeof(prach_cfg),'
                   ' sizeof(prach_result), sizeof(prach_ue_log), sizeof(prach_timing)); return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = list(map(int, subprocess.check_output([str(exe)]).split()))
    assert sizes == [C.sizeof(pkg.PrachCfg), C.sizeof(pkg.PrachResult), C.sizeof(pkg.PrachUeLog), C.sizeof(pkg.PrachTiming)]


def test_no_device_fails_loudly(pkg):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PrachError) as ei:
        pkg.Engine(0)
    assert ei.value.status == -3
    p = subprocess.run([pkg.CLI_PATH, "--nue", "100"], capture_output=True, text=True)
    assert p.returncode == 2 and "no CPU fallback" in p.stderr


def test_product_does_not_reference_oracle():
    """The product path must never route through oracle/ (it is the checker, not the product)."""
    for top in ("5g-nr-randomaccess_amd", "scripts", "include"):
      for dp, _, fns in os.walk(os.path.join(ROOT, top)):
        for fn in fns:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cpp")) or fn == "Makefile":
                text = open(os.path.join(dp, fn), errors="ignore").read()
                code = "\n".join(l for l in text.split("\n") if not l.lstrip().startswith(("//", "*", "/*", "#")))
                assert "liboracle" not in code and "from oracle" not in code and "import oracle" not in code, fn
