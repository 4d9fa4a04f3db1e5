#!/usr/bin/env python3
"""Regenerate the Fortran binding block of INTEGRATION.md from bindings/fortran/isca_dyn_c.F90 (derived types + interface block: the
part a maintainer copies), between the BEGIN/END markers.  tests/test_host_cpu.py checks that the document is in sync.
usage: python tools/gen_integration_snippet.py [--check]"""
import os, re, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN generated from bindings/fortran/isca_dyn_c.F90 -->", "<!-- END generated -->"


def snippet():
    src = open(os.path.join(REPO, "bindings", "fortran", "isca_dyn_c.F90")).read()
    body = src[src.index("module isca_dyn_c"):src.index("\ncontains")]
    return "```fortran\n" + body.rstrip() + "\n! (contains: isca_message(), check_abi() -- see the file)\nend module isca_dyn_c\n```"


def render(doc):
    a, b = doc.index(BEGIN), doc.index(END)
    return doc[:a + len(BEGIN)] + "\n" + snippet() + "\n" + doc[b:]


if __name__ == "__main__":
    path = os.path.join(REPO, "INTEGRATION.md")
    doc = open(path).read()
    new = render(doc)
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else 1)
    open(path, "w").write(new)
