"""Undefined global names in python sources (no pyflakes in this image): every name a function reads as a global must be defined at
module level (import, def, class, assignment) or be a builtin.   python tools/check_names.py video-gcp_amd/*.py"""
import ast, builtins, symtable, sys

def module_names(tree):
    names = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                names.add((a.asname or a.name).split(".")[0])
    for n in tree.body:
        if isinstance(n, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            names.add(n.name)
        for t in ast.walk(n) if isinstance(n, (ast.Assign, ast.AugAssign, ast.AnnAssign, ast.For, ast.With, ast.If, ast.Try)) else []:
            if isinstance(t, ast.Name) and isinstance(t.ctx, ast.Store):
                names.add(t.id)
    return names

def walk(tab, known, path, bad):
    for s in tab.get_symbols():
        if s.is_global() and s.is_referenced() and s.get_name() not in known:
            bad.append((path, tab.get_name(), tab.get_lineno(), s.get_name()))
    for c in tab.get_children():
        walk(c, known, path, bad)

bad = []
for p in sys.argv[1:]:
    src = open(p).read()
    known = module_names(ast.parse(src))
    walk(symtable.symtable(src, p, "exec"), known, p, bad)
for b in bad:
    print("%s: in %s (line %d): undefined name %s" % b)
print("undefined names:", len(bad))
sys.exit(1 if bad else 0)
