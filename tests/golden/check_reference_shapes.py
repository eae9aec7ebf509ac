#!/usr/bin/env python3
"""Development aid (build container only): tests/reference_shapes.py against the plans the reference's own text lowers to — prints,
per query, whether the name-free plans agree, and both of them when they do not.  Reads /root/reference at run time; stores nothing."""
import ast
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from sdqlpy_amd import frontend  # noqa: E402
import reference_shapes  # noqa: E402

src = open("/root/reference/test/test_all.py").read()
tree, lines = ast.parse(src), src.splitlines()
only = sys.argv[1:]
for node in tree.body:
    if isinstance(node, ast.FunctionDef) and node.name in reference_shapes.QUERIES and (not only or node.name in only):
        first = min([d.lineno for d in node.decorator_list] + [node.lineno])
        ref = frontend.lower_source("\n".join(lines[first - 1:node.end_lineno]), node.name, first).fingerprint()
        own = frontend.lower_function(reference_shapes.QUERIES[node.name]).fingerprint()
        print(node.name, "same plan" if ref == own else "DIFFERENT")
        if ref != own:
            for a, b in zip(ref.splitlines(), own.splitlines()):
                if a != b:
                    print("   reference:", a)
                    print("   own      :", b)
            if len(ref.splitlines()) != len(own.splitlines()):
                print("   line counts", len(ref.splitlines()), len(own.splitlines()))
