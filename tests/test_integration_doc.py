"""INTEGRATION.md is the reference-side binding a maintainer would follow: its C snippets must compile against
include/deeprl_amd.h and its Python snippets must call the exports with the header's argument counts (round 3 shipped
a snippet with `dra_comm_init_rank`'s arguments in the wrong order: VERDICT r3)."""
import ast
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")


def _blocks(lang):
    src = open(DOC).read()
    return re.findall(r"```%s\n(.*?)```" % lang, src, flags=re.S)


def test_c_snippets_compile_against_the_header(tmp_path):
    blocks = _blocks("c")
    assert blocks, "INTEGRATION.md shows at least one C binding"
    for i, b in enumerate(blocks):
        f = tmp_path / ("snippet%d.c" % i)
        f.write_text(b)
        r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter",
                            "-I", os.path.join(ROOT, "include"), str(f)], capture_output=True, text=True)
        assert r.returncode == 0, "C snippet %d of INTEGRATION.md does not compile:\n%s\n%s" % (i, b, r.stderr)


def test_python_snippets_call_exports_with_the_headers_arity():
    from deeprl_amd._lib import parse_header
    protos = parse_header()
    blocks = _blocks("python")
    assert blocks
    n_calls = 0
    for b in blocks:
        tree = ast.parse(b)                     # syntax
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("dra_"):
                name = node.func.attr
                assert name in protos, "%s is not declared in include/deeprl_amd.h" % name
                assert len(node.args) == len(protos[name]), "%s: %d arguments in INTEGRATION.md, %d in the header" % (
                    name, len(node.args), len(protos[name]))
                n_calls += 1
    assert n_calls >= 3


def test_every_export_named_in_the_doc_exists():
    from deeprl_amd._lib import parse_header
    protos = parse_header()
    src = open(DOC).read()
    for name in set(re.findall(r"\b(dra_[a-z0-9_]+)\b", src)):
        if name in protos or name.rstrip("_") != name:
            continue
        # struct / typedef names and prefixes written with a trailing wildcard are not exports
        hdr = open(os.path.join(ROOT, "include", "deeprl_amd.h")).read()
        assert re.search(r"\b%s\b" % re.escape(name), hdr) or any(k.startswith(name) for k in protos), name
