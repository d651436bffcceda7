"""A C program (tests/c_abi/abi_check.c) against include/block_aligner_hip.h: compiles and links with plain gcc
everywhere; on a GPU box it runs and its answers are compared with the oracle."""
import os
import subprocess

import pytest

from block_aligner_amd import scores as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "block_aligner_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "abi_check")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "abi_check.c"), "-o", exe,
                           "-L", LIBDIR, "-lblock_aligner_hip", f"-Wl,-rpath,{LIBDIR}"])
    return exe


def test_c_program_compiles_and_links(tmp_path):
    exe = build(tmp_path)
    out = subprocess.check_output([exe, "--link-only"], text=True)
    assert "sizeof AlignResult=24 OpLen=16 Gaps=2 SizeRange=16" in out      # c/block_aligner.h:90-126 layouts
    assert "percent_len 128 1024" in out                                     # lib.rs:109-111
    assert "operation 2 1" in out                                            # `enum Operation` / one-byte `Operation` (c/block_aligner.h:17-57)


@pytest.mark.gpu
def test_c_program_matches_oracle(tmp_path, oracle):
    exe = build(tmp_path)
    lines = subprocess.check_output([exe], text=True, timeout=300).splitlines()
    got = {l.split()[0]: l.split()[1:] for l in lines}

    def expect(name, res, cigar):
        assert got[name] == [str(res["score"]), str(res["query_idx"]), str(res["reference_idx"]), cigar or "-"], (name, got[name], res)

    g = (-11, -1)
    r = oracle.align(S.BLOSUM62, b"AAAAAAAA", b"AARAAAA", g, (32, 32), 0, ())
    expect("aa_global", r, None)
    r = oracle.align(S.BLOSUM62, b"AAAAAAAA", b"AARAAAA", g, (32, 32), 0, ("trace",))
    expect("aa_trace", r, r["cigar"])
    r = oracle.align(S.BLOSUM62, b"MKVLAARNDCEQGHILKMFPSTWYV", b"MKVLAARNDCEQGHILKMFPSTWYVAAAAAAAA", g, (16, 64), 50, ("x_drop",))
    expect("aa_xdrop", r, None)
    r = oracle.align(S.static_matrix("BLOSUM50"), b"MKVLAARNDCEQGHILKMFPSTWYV", b"MKVLARNDCEQGHILKMMFPSTWYV", g, (16, 64), 50, ("trace", "x_drop"))
    expect("aa_trace_xdrop", r, r["cigar"])
    p = S.AAProfile.from_bytes(b"ARNDCEQGHIKARNDCEQGHI", 32, 2, -1, -3, 0, -3, -1)
    r = oracle.align_profile(b"ARNDCEQGHIARNDCEQGHI", p, (32, 32), 0, ("trace",))
    expect("profile", r, r["cigar"])
    pairs = [(b"TTTTTTTTAAAAAAATTTTTTTTT", b"TTAAAAAAATTTTTTTTTTTT"), (b"ACGTACGTACGTTTACGTACGT", b"ACGTACGTACGTACGTACGT"), (b"", b"ACGT")]
    for k, (q, rr) in enumerate(pairs):
        r = oracle.align(S.NW1, q, rr, (-2, -1), (32, 256), 0, ("trace",), cigar_eq=True)
        expect(f"batch{k}", r, r["cigar"])
    assert got["batch0"][:3] == ["7", "24", "21"] and got["batch0"][3] == "2=6I16=3D"    # README.md:44-53
