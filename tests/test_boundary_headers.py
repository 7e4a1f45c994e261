"""Boundary guard (pins nothing about behaviour): the reference's own headers for the hot path compile UNCHANGED against
include/pkmn.h -- the claim INTEGRATION.md makes.  Needs the reference checkout (skipped where it is absent, e.g. on the
GPU box); -fsyntax-only, so nothing from the reference is built or linked."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/cpp/include"
HEADERS = ["libpkmn/pkmn.h", "libpkmn/init.h", "libpkmn/data.h", "libpkmn/layout.h", "search/durations.h", "encode/battle/battle.h",
           "encode/battle/policy.h", "search/poke-engine-evaluate.h", "train/battle/compressed-frame.h"]


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="reference checkout or g++ absent")
@pytest.mark.parametrize("header", HEADERS)
def test_reference_header_compiles_against_our_pkmn_h(header, tmp_path):
    src = tmp_path / "tu.cc"
    # <pkmn.h> (the generated libpkmn header the reference includes) resolves to include/pkmn.h: -I include comes first
    # policy.h relies on its includer having pulled in the battle views first (as the reference's own translation units do)
    pre = "#include <libpkmn/data.h>\n" if header == "encode/battle/policy.h" else ""
    src.write_text("#include <array>\n#include <cassert>\n#include <cstring>\n#include <string>\n#include <vector>\n%s#include <%s>\nint main() { return 0; }\n" % (pre, header))
    r = subprocess.run(["g++", "-std=c++2b", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), "-I", REF, str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
