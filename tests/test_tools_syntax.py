"""The scripts under tools/ are the lab notebook the profiles and DESIGN.md cite (run on the GPU box by hand). None of them is part of the product;
this keeps them loadable: every Python file compiles, every shell script parses, and the development switches they set (BA_* variables) are ones
the development build of the host library still reads."""
import os
import py_compile
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(ROOT, "tools")) for f in fs if "__pycache__" not in d]


@pytest.mark.parametrize("path", sorted(p for p in TOOLS if p.endswith(".py")), ids=lambda p: os.path.relpath(p, ROOT))
def test_python_tools_compile(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", sorted(p for p in TOOLS if p.endswith(".sh")), ids=lambda p: os.path.relpath(p, ROOT))
def test_shell_tools_parse(path):
    subprocess.run(["bash", "-n", path], check=True)


def test_switches_set_by_the_tools_are_read_by_the_development_build():
    host = open(os.path.join(ROOT, "block_aligner_amd", "csrc", "ba_host.cpp")).read()
    known = set(re.findall(r'dev_env\("(BA_[A-Z0-9_]+)"\)', host))
    # variables of the tools themselves (library choice, generators, sweeps), not of the library
    own = {"BA_LIB", "BA_GEN_WORKERS", "BA_HIP_LIB"}
    unknown = {}
    for p in TOOLS:
        if not p.endswith((".py", ".sh")):
            continue
        for v in set(re.findall(r"\b(BA_[A-Z0-9_]+)=", open(p).read())):
            if v not in known and v not in own:
                unknown.setdefault(v, []).append(os.path.relpath(p, ROOT))
    assert not unknown, unknown
