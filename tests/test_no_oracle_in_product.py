"""The product path must never import, call or fall back to the oracle (or /root/reference)."""
import ast
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _code_strings(src: str):
    """string constants of a module that are NOT docstrings (citations of reference file:line live in docstrings
    and comments; a path the code could open, import from or put on sys.path would be a constant)"""
    tree = ast.parse(src)
    doc = set()
    for node in ast.walk(tree):
        if isinstance(node, (ast.Module, ast.ClassDef, ast.FunctionDef, ast.AsyncFunctionDef)) and node.body:
            first = node.body[0]
            if isinstance(first, ast.Expr) and isinstance(first.value, ast.Constant) and isinstance(first.value.value, str):
                doc.add(id(first.value))
    return [n.value for n in ast.walk(tree) if isinstance(n, ast.Constant) and isinstance(n.value, str) and id(n) not in doc]


def test_product_sources_do_not_touch_oracle_or_reference():
    bad = []
    for p in list((ROOT / "consistentid_amd").rglob("*.py")) + list((ROOT / "consistentid_amd" / "csrc").glob("*")):
        txt = p.read_text(errors="ignore")
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "import_module(\"oracle" in txt:
            bad.append(str(p))
        if p.suffix == ".py" and any("/root/reference" in c for c in _code_strings(txt)):
            bad.append(str(p) + " (names /root/reference outside a docstring: only citations may)")
    assert not bad, bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    txt = (ROOT / "bench.py").read_text()
    for m in re.finditer(r"^(\s*)(from|import)\s+oracle\b.*$", txt, re.M):
        assert len(m.group(1)) > 0, "oracle import must be local to the cpu_baseline function"
    assert "def cpu_baseline" in txt
