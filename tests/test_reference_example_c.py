"""The reference's own C example (c/example.c: three alignments through the 54-function C API) compiles and links UNCHANGED against
include/block_aligner_hip.h + libblock_aligner_hip.so ("example.c becomes a smoke test of our C-ABI lib").
The source is read where it lies under /root/reference (never copied into the repo); the only thing added is a one-line
`block_aligner.h` that includes this repo's header. Skipped where the reference tree is absent (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "block_aligner_amd", "lib")
EXAMPLE = "/root/reference/c/example.c"


@pytest.mark.skipif(not os.path.exists(EXAMPLE), reason="the reference tree is not present on this machine")
def test_reference_example_compiles_and_links_unchanged(tmp_path):
    (tmp_path / "block_aligner.h").write_text('#include "block_aligner_hip.h"\n')
    exe = str(tmp_path / "example")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", str(tmp_path), "-I", os.path.join(ROOT, "include"), EXAMPLE, "-o", exe,
                           "-L", LIBDIR, "-lblock_aligner_hip", f"-Wl,-rpath,{LIBDIR}"])
    # every block_* symbol the example calls resolves against the library (the link above would have failed otherwise); the three
    # expected answers -- score 12 twice and the profile alignment's -- are checked on the GPU by tests/test_c_abi.py's C caller,
    # which runs the same calls (c/example.c:8-33,41-81,90-125)
    syms = subprocess.check_output(["nm", "-u", exe], text=True)
    used = sorted({l.split()[-1].split("@")[0] for l in syms.splitlines() if "block_" in l})
    assert "block_align_aa" in used and "block_align_aa_trace" in used and "block_align_profile_aa" in used, used
