"""The product path must never import, call or fall back to the oracle (or /root/reference)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_product_sources_do_not_touch_oracle_or_reference():
    bad = []
    for p in list((ROOT / "consistentid_amd").rglob("*.py")) + list((ROOT / "consistentid_amd" / "csrc").glob("*")):
        txt = p.read_text(errors="ignore")
        if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "import_module(\"oracle" in txt:
            bad.append(str(p))
        if "/root/reference" in txt and p.suffix == ".py" and re.search(r"open\(|sys\.path|importlib", txt):
            bad.append(str(p) + " (reads /root/reference)")
    assert not bad, bad


def test_bench_uses_oracle_only_in_cpu_baseline():
    txt = (ROOT / "bench.py").read_text()
    for m in re.finditer(r"^(\s*)(from|import)\s+oracle\b.*$", txt, re.M):
        assert len(m.group(1)) > 0, "oracle import must be local to the cpu_baseline function"
    assert "def cpu_baseline" in txt
