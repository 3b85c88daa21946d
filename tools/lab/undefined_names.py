"""Poor man's pyflakes (the image has no linter): report Name loads that no enclosing scope binds.
    python -m tools.lab.undefined_names file.py ..."""
import ast
import builtins
import sys


def bound_names(node):
    """Names bound directly in the scope `node` opens (not in nested function / class scopes)."""
    out = set()

    def visit(n, top=False):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef, ast.Lambda)) and not top:
            if not isinstance(n, ast.Lambda):
                out.add(n.name)
            return
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                out.add(arg.arg)
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        if isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        if isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        if isinstance(n, (ast.Global, ast.Nonlocal)):
            out.update(n.names)
        if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
            pass                                    # comprehension targets: treated as bound in the enclosing scope
        for c in ast.iter_child_nodes(n):
            visit(c)
    visit(node, top=True)
    return out


def check(path):
    tree = ast.parse(open(path).read(), path)
    problems = []

    def walk(node, scopes):
        scopes = scopes + [bound_names(node)]
        for n in ast.iter_child_nodes(node):
            inner(n, scopes)

    def inner(n, scopes):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            for d in getattr(n, "decorator_list", []):
                inner(d, scopes)
            for d in n.args.defaults + [k for k in n.args.kw_defaults if k is not None]:
                inner(d, scopes)
            walk(n, scopes)
            return
        if isinstance(n, ast.ClassDef):
            for b in n.bases + n.decorator_list:
                inner(b, scopes)
            # class body names are visible only in the body itself, not in methods
            body_scope = bound_names(n)
            for c in n.body:
                if isinstance(c, (ast.FunctionDef, ast.AsyncFunctionDef)):
                    inner(c, scopes)
                else:
                    inner(c, scopes + [body_scope])
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
            if not any(n.id in s for s in scopes) and not hasattr(builtins, n.id):
                problems.append((n.lineno, n.id))
        for c in ast.iter_child_nodes(n):
            inner(c, scopes)
    walk(tree, [])
    return problems


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for line, name in check(p):
            print("%s:%d: undefined name %r" % (p, line, name))
            bad += 1
    sys.exit(1 if bad else 0)
