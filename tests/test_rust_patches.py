"""rust/ -- the `simd_hip` feature for the reference crate -- is source nobody here can compile (no rustc / cargo in the image). What CAN be
checked without a toolchain is checked here, on a copy of /root/reference (skipped where the reference is absent, e.g. on the GPU box):
  * every patch in rust/ applies with `patch -p1`, no rejects, no fuzz;
  * with only `simd_hip` enabled, no item that survives the cfg gates mentions the SIMD layer (Simd, HalfSimd, LutSimd, TraceType, the
    simd_* / halfsimd_* intrinsics wrappers, the lane count L): a small cfg-aware scanner removes what `#[cfg(not(feature = "simd_hip"))]`,
    the CPU feature gates and `#[cfg(test)]` compile out and greps the rest;
  * the FFI block of src/hip.rs declares only functions that include/block_aligner_hip.h declares.
"""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
RUST = os.path.join(ROOT, "rust")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")) or shutil.which("patch") is None,
                                reason="needs /root/reference and patch(1)")

CPU_FEATURES = ("simd_sse2", "simd_avx2", "simd_wasm", "simd_neon")
SIMD_WORDS = re.compile(r"\b(Simd|HalfSimd|LutSimd|TraceType|L|simd_[a-z0-9_]+|halfsimd_[a-z0-9_]+|lutsimd_[a-z0-9_]+)\b")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    d = tmp_path_factory.mktemp("crate")
    shutil.copy(os.path.join(REF, "Cargo.toml"), d)
    shutil.copytree(os.path.join(REF, "src"), d / "src")
    os.makedirs(d / "examples")
    patches = sorted(f for f in os.listdir(RUST) if f.endswith(".patch"))
    assert patches == ["Cargo.toml.patch", "cigar.rs.patch", "lib.rs.patch", "scan_block.rs.patch", "scores.rs.patch"]
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(RUST, p)], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, (p, r.stdout, r.stderr)
        assert "fuzz" not in r.stdout and "offset" not in r.stdout and "FAILED" not in r.stdout, (p, r.stdout)
    assert not [f for _, _, fs in os.walk(d) for f in fs if f.endswith((".rej", ".orig"))]
    for f in os.listdir(os.path.join(RUST, "src")):
        shutil.copy(os.path.join(RUST, "src", f), d / "src")
    shutil.copy(os.path.join(RUST, "examples", "dump_golden.rs"), d / "examples")
    return d


def strip_comments(text):
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def cfg_enabled(expr):
    """Value of a cfg predicate in a build with ONLY the simd_hip feature (and not under test)."""
    expr = expr.strip()
    m = re.fullmatch(r'feature\s*=\s*"([a-z0-9_]+)"', expr)
    if m:
        return m.group(1) == "simd_hip"
    if expr == "test":
        return False
    if expr.startswith("target_arch"):
        return True   # (irrelevant to the question asked here)
    for op in ("not", "any", "all"):
        if expr.startswith(op + "(") and expr.endswith(")"):
            inner, parts, depth, cur = expr[len(op) + 1:-1], [], 0, ""
            for ch in inner:
                if ch == "," and depth == 0:
                    parts.append(cur); cur = ""
                else:
                    depth += ch == "("; depth -= ch == ")"; cur += ch
            if cur.strip():
                parts.append(cur)
            vals = [cfg_enabled(p) for p in parts]
            return (not vals[0]) if op == "not" else (any(vals) if op == "any" else all(vals))
    raise AssertionError("cfg predicate not understood: " + expr)


def item_end(text, start):
    """Index just past the item that begins at `start` (after its attributes). The item's header ends with the first line that ends in
    `;` (no body), in `{` (the body's brace: const-generic arguments like `Block<{ TRACE }>` earlier in the header are not it) or in `}`
    (a one-line body)."""
    i = start
    while True:
        nl = text.find("\n", i)
        nl = len(text) if nl < 0 else nl
        line = text[i:nl].rstrip()
        if line.endswith(";") and line.count("{") == line.count("}"):
            return nl
        if line.endswith("{"):
            depth, k = 1, i + len(line)
            while depth:
                ch = text[k]
                depth += ch == "{"; depth -= ch == "}"
                k += 1
            return k
        if line.endswith("}") and "{" in line:
            return nl
        assert nl < len(text), "unterminated item"
        i = nl + 1


def surviving(text):
    """The source with every cfg-disabled item removed (a build with only `simd_hip`)."""
    text = strip_comments(text)
    out, i = [], 0
    attr = re.compile(r"#\[cfg\((.*)\)\]\s*$")
    lines = text.split("\n")
    pos = [0]
    for ln in lines:
        pos.append(pos[-1] + len(ln) + 1)
    k = 0
    keep_from = 0
    while k < len(lines):
        m = attr.match(lines[k].strip()) if lines[k].strip().startswith("#[cfg(") else None
        if m and not cfg_enabled(m.group(1)):
            # skip this attribute, any further attributes, then the item
            j = k + 1
            while j < len(lines) and (lines[j].strip().startswith("#[") or not lines[j].strip()):
                j += 1
            end = item_end(text, pos[j])
            out.append(text[keep_from:pos[k]])
            keep_from = end
            while k < len(lines) and pos[k + 1] <= end:
                k += 1
            if pos[k] < end:   # the item ended inside this line
                k += 1
            continue
        k += 1
    out.append(text[keep_from:])
    return "".join(out)


def test_patches_apply_cleanly(patched):
    lib = (patched / "src" / "lib.rs").read_text()
    assert 'feature = "simd_hip"' in lib and "pub mod hip;" in lib
    assert "simd_hip = []" in (patched / "Cargo.toml").read_text()
    assert "fn set_runs" in (patched / "src" / "cigar.rs").read_text()


def test_no_simd_identifier_survives_a_simd_hip_only_build(patched):
    lib = surviving((patched / "src" / "lib.rs").read_text())
    mods = re.findall(r"pub mod (\w+);", lib)
    assert set(mods) == {"hip", "scan_block", "scores", "cigar"}, mods   # (no avx2 / sse2 / neon / simd128 / ffi)
    assert "compile_error!" not in lib
    for name in ("scan_block.rs", "scores.rs", "cigar.rs", "hip.rs", "scan_block_hip.rs"):
        code = surviving((patched / "src" / name).read_text())
        code = re.sub(r'"(\\.|[^"\\])*"', '""', code)   # string literals (the feature names among them)
        hits = sorted({m.group(0) for m in SIMD_WORDS.finditer(code)})
        assert not hits, (name, hits)
    sb = surviving((patched / "src" / "scan_block.rs").read_text())
    for kept in ("pub struct PaddedBytes", "pub struct AlignResult", "pub struct Rectangle", "pub use hip_impl::{Block, Trace};"):
        assert kept in sb, kept
    for gone in ("macro_rules!", "struct Allocated", "struct Aligned", "mod tests", "enum Direction"):
        assert gone not in sb, gone
    sc = surviving((patched / "src" / "scores.rs").read_text())
    assert sc.count("const HIP_KIND: i32") == 4 and "fn hip_raw" in sc and "get_scores" not in sc and "get_gap_open_right_C" not in sc


def test_the_cpu_build_is_unchanged_by_the_gates(patched):
    """With a CPU feature and without simd_hip every original line is still there: the patches only add lines (and widen five cfg gates)."""
    for name in ("scan_block.rs", "scores.rs", "cigar.rs"):
        ref = open(os.path.join(REF, "src", name)).read().split("\n")
        new = (patched / "src" / name).read_text().split("\n")
        it = iter(new)
        assert all(any(x == line for x in it) for line in ref), name   # ref is a subsequence of new


def test_ffi_declarations_exist_in_the_header(patched):
    hip_rs = strip_comments((patched / "src" / "hip.rs").read_text())
    block = hip_rs[hip_rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    fns = re.findall(r"pub fn (\w+)\(", block)
    assert len(fns) > 50
    header = open(os.path.join(ROOT, "include", "block_aligner_hip.h")).read()
    missing = [f for f in fns if not re.search(r"\b%s\s*\(" % f, header)]
    assert not missing, missing
    ex = (patched / "examples" / "dump_golden.rs").read_text()
    assert "Block::<TRACE, X_DROP, LOCAL_START, FQS, FQE>::new" in ex and "crate_golden_input.tsv" in ex
