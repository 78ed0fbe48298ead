"""The product package must never import, call or fall back to the CPU oracle (or to the reference)."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "object_detection_cib_amd")


def _imports(path):
    tree = ast.parse(open(path).read(), path)
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name
        elif isinstance(node, ast.ImportFrom) and node.level == 0 and node.module:
            yield node.module


def test_package_does_not_import_oracle_or_reference():
    bad = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                p = os.path.join(dirpath, f)
                for mod in _imports(p):
                    root = mod.split(".")[0]
                    if root in ("oracle", "kod", "tests", "torchvision"):
                        bad.append((os.path.relpath(p, ROOT), mod))
                src = open(p).read()
                assert "/root/reference" not in src, p
    assert not bad, bad


def test_only_checkers_import_oracle():
    allowed = {"tests", "oracle", "tools"}
    offenders = []
    for name in os.listdir(ROOT):
        p = os.path.join(ROOT, name)
        if name.endswith(".py") and name not in ("bench.py", "__graft_entry__.py"):
            if any(m.split(".")[0] == "oracle" for m in _imports(p)):
                offenders.append(name)
        elif os.path.isdir(p) and name not in allowed and not name.startswith("."):
            for dirpath, _, files in os.walk(p):
                for f in files:
                    if f.endswith(".py") and any(m.split(".")[0] == "oracle" for m in _imports(os.path.join(dirpath, f))):
                        offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_bench_uses_oracle_only_for_inputs_and_cpu_baseline():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    users = {}
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.split(".")[0] == "oracle":
                users.setdefault(fn.name, set()).update(a.name for a in node.names)
    assert set(users) <= {"cpu_baseline", "synth_batch"}, users
    assert users.get("synth_batch", set()) <= {"synth"}, users


def test_bench_touches_oracle_only_in_its_cpu_baseline_leg():
    """bench.py may time the oracle as the CPU baseline, nothing else: every `oracle` import sits inside
    cpu_baseline(); __graft_entry__ only inside build() (compiling / importing the checker) and smoke()."""
    for fname, ok_funcs in (("bench.py", {"cpu_baseline"}), ("__graft_entry__.py", {"build", "smoke"})):
        tree = ast.parse(open(os.path.join(ROOT, fname)).read(), fname)
        inside = set()
        for fn in ast.walk(tree):
            if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef)) and fn.name in ok_funcs:
                for node in ast.walk(fn):
                    inside.add(id(node))
        for node in ast.walk(tree):
            mods = []
            if isinstance(node, ast.Import):
                mods = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom) and node.module:
                mods = [node.module]
            if any(m.split(".")[0] == "oracle" for m in mods):
                assert id(node) in inside, (fname, getattr(node, "lineno", None))
