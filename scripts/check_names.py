#!/usr/bin/env python3
"""Undefined global names in the package's modules (no linter in this image): every name a function reads as a global must be
bound at module level or be a builtin.  The GPU-only paths of the engine are never executed by the CPU suite, so a missing
import there would otherwise first show on the GPU box.

    python scripts/check_names.py [files...]        exit code 1 when something is unbound
"""
import ast
import builtins
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def module_bindings(tree):
    names = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                names.add((a.asname or a.name).split(".")[0])
    for node in tree.body:
        for sub in ast.walk(node) if not isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) else [node]:
            if isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(sub.name)
            elif isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                names.add(sub.id)
    return names


class Scope(ast.NodeVisitor):
    def __init__(self, module_names, report, fname):
        self.module_names, self.report, self.fname = module_names, report, fname
        self.stack = []

    def _locals_of(self, node):
        loc = set()
        args = node.args
        for a in args.posonlyargs + args.args + args.kwonlyargs + ([args.vararg] if args.vararg else []) + ([args.kwarg] if args.kwarg else []):
            loc.add(a.arg)
        glob = set()
        for sub in ast.walk(node):
            if sub is not node and isinstance(sub, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                loc.add(sub.name)
            if isinstance(sub, ast.Lambda):
                for a in sub.args.posonlyargs + sub.args.args + sub.args.kwonlyargs + ([sub.args.vararg] if sub.args.vararg else []) + ([sub.args.kwarg] if sub.args.kwarg else []):
                    loc.add(a.arg)
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, (ast.Store, ast.Del)):
                loc.add(sub.id)
            if isinstance(sub, (ast.Import, ast.ImportFrom)):
                for a in sub.names:
                    loc.add((a.asname or a.name).split(".")[0])
            if isinstance(sub, ast.ExceptHandler) and sub.name:
                loc.add(sub.name)
            if isinstance(sub, ast.Global):
                glob.update(sub.names)
            if isinstance(sub, ast.arg):
                loc.add(sub.arg)
        return loc - glob

    def visit_FunctionDef(self, node):
        self.stack.append(self._locals_of(node))
        known = set().union(*self.stack) | self.module_names | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        decorators = {id(n) for dec in node.decorator_list for n in ast.walk(dec)}      # (`@prop.setter` names a class-level binding)
        for sub in ast.walk(node):
            if id(sub) in decorators:
                continue
            if isinstance(sub, ast.Name) and isinstance(sub.ctx, ast.Load) and sub.id not in known:
                self.report.append(f"{self.fname}:{sub.lineno}: {sub.id!r} is not bound (function {node.name})")
        self.stack.pop()          # (nested functions were covered by the walk above with the union of their own locals)

    visit_AsyncFunctionDef = visit_FunctionDef

    def visit_ClassDef(self, node):
        class_names = {n.name for n in node.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
        for sub in node.body:
            if isinstance(sub, ast.Assign):
                for t in sub.targets:
                    for nm in ast.walk(t):
                        if isinstance(nm, ast.Name):
                            class_names.add(nm.id)
        self.stack.append(set())          # class-level names are NOT visible inside methods
        for sub in node.body:
            self.visit(sub)
        self.stack.pop()


def check(path):
    tree = ast.parse(open(path).read(), path)
    report = []
    Scope(module_bindings(tree), report, os.path.relpath(path, ROOT)).visit(tree)
    return report


def main(argv):
    files = argv or sorted(glob.glob(os.path.join(ROOT, "lanpaint_amd", "*.py")) + glob.glob(os.path.join(ROOT, "benchkit", "*.py"))
                           + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])
    bad = [ln for f in files for ln in check(f)]
    print("\n".join(bad) if bad else f"{len(files)} files: every global name is bound")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
