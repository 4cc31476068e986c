"""The Rust side of the [patch] integration (integration/rust/) has never met a compiler — there is none in this image.  What CAN be held here, on the CPU: the sources are
lexically sound (delimiters balance outside strings, comments, char literals and lifetimes), and every `sys::zkhip_*` call in the two shims names a function that the generated
bindings (zkhip-sys/src/lib.rs, itself checked against include/zkhip.h by test_abi.py) declare, with the declared number of arguments — so a change of the C ABI cannot silently
strand the shim a maintainer would patch into halo2curves / halo2_proofs (/root/reference/Cargo.toml:14-36 for the dependency spellings they replace)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUST = sorted(glob.glob(os.path.join(ROOT, "integration", "rust", "*", "src", "*.rs")) + glob.glob(os.path.join(ROOT, "integration", "rust", "*", "build.rs")))


def strip(src):
    """Rust source with comments, string / raw-string / char literals blanked (same length), lifetimes left alone"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if src.startswith("//", i):
            j = src.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i)); i = j
        elif src.startswith("/*", i):
            depth, j = 1, i + 2
            while j < n and depth:
                if src.startswith("/*", j): depth += 1; j += 2
                elif src.startswith("*/", j): depth -= 1; j += 2
                else: j += 1
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == "r" and re.match(r'r#*"', src[i:]):
            m = re.match(r'r(#*)"', src[i:])
            end = src.find('"' + m.group(1), i + len(m.group(0)))
            assert end >= 0, "unterminated raw string"
            j = end + 1 + len(m.group(1))
            out.append("".join(ch if ch == "\n" else " " for ch in src[i:j])); i = j
        elif c == '"':
            j = i + 1
            while j < n and src[j] != '"':
                j += 2 if src[j] == "\\" else 1
            assert j < n, "unterminated string literal"
            out.append('"' + "".join(ch if ch == "\n" else " " for ch in src[i + 1:j]) + '"'); i = j + 1
        elif c == "'":
            m = re.match(r"'(\\.[^']*|[^'\\])'", src[i:])
            if m:      # a char literal
                out.append(" " * len(m.group(0))); i += len(m.group(0))
            else:      # a lifetime or a label
                out.append(c); i += 1
        else:
            out.append(c); i += 1
    return "".join(out)


def test_sources_exist_and_balance():
    assert len(RUST) >= 5, RUST
    pairs = {")": "(", "]": "[", "}": "{"}
    for f in RUST:
        text = strip(open(f).read())
        stack = []
        for pos, ch in enumerate(text):
            if ch in "([{":
                stack.append((ch, pos))
            elif ch in ")]}":
                assert stack and stack[-1][0] == pairs[ch], f"{os.path.relpath(f, ROOT)}: unbalanced '{ch}' at line {text.count(chr(10), 0, pos) + 1}"
                stack.pop()
        assert not stack, f"{os.path.relpath(f, ROOT)}: unclosed '{stack[-1][0]}' from line {text.count(chr(10), 0, stack[-1][1]) + 1}"


def _top_level_args(text, open_paren):
    """number of top-level comma-separated arguments of the call whose '(' is at open_paren"""
    depth, args, seen = 0, 0, False
    for pos in range(open_paren, len(text)):
        ch = text[pos]
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
            if depth == 0:
                return args + (1 if seen else 0)
        elif depth == 1:
            if ch == ",":
                args += 1; seen = False
            elif not ch.isspace():
                seen = True
    raise AssertionError("unterminated call")


def test_every_shim_call_matches_the_generated_bindings():
    sys_rs = strip(open(os.path.join(ROOT, "integration", "rust", "zkhip-sys", "src", "lib.rs")).read())
    declared = {}
    for m in re.finditer(r"pub fn (zkhip_\w+)\s*\(", sys_rs):
        declared[m.group(1)] = _top_level_args(sys_rs, m.end() - 1)
    assert len(declared) >= 100
    calls = 0
    for f in RUST:
        if f.endswith(os.path.join("zkhip-sys", "src", "lib.rs")):
            continue
        text = strip(open(f).read())
        for m in re.finditer(r"\bsys::(zkhip_\w+)\s*\(", text):
            name = m.group(1)
            assert name in declared, f"{os.path.relpath(f, ROOT)}: sys::{name} is not declared by zkhip-sys (include/zkhip.h)"
            got = _top_level_args(text, m.end() - 1)
            assert got == declared[name], f"{os.path.relpath(f, ROOT)} line {text.count(chr(10), 0, m.start()) + 1}: sys::{name} called with {got} arguments, declared with {declared[name]}"
            calls += 1
    assert calls >= 10, calls
