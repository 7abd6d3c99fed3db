"""CPU: the documents point at things that exist.  Every `profiles/...`, `scripts/...`, `tests/...` path that DESIGN.md, README.md,
INTEGRATION.md or profiles/r05_README.txt name is in the tree (globs allowed), and profiles/kernel_traffic.json says which csrc tree it
was measured on (bench.py reports `traffic: null` + a `traffic_stale` note when the library was built from another one)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths(text):
    for m in re.finditer(r"`((?:profiles|scripts|tests|oracle|include|vistaocr_amd)/[A-Za-z0-9_./*\-]+)`", text):
        p = m.group(1).rstrip(".")
        if "::" in p:
            p = p.split("::")[0]
        yield p


def test_documents_name_existing_files():
    missing = []
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        for p in _paths(open(os.path.join(ROOT, doc)).read()):
            if "rNN" in p or p.endswith("/"):
                continue
            if not glob.glob(os.path.join(ROOT, p)):
                missing.append((doc, p))
    assert not missing, missing


def test_kernel_traffic_names_its_csrc_tree():
    tj = json.load(open(os.path.join(ROOT, "profiles", "kernel_traffic.json")))
    assert re.fullmatch(r"[0-9a-f]{64}", tj.get("csrc_tree_hash", ""))
    fams = [k for k, v in tj.items() if isinstance(v, dict)]
    assert len(fams) >= 4 and all("fetch_KB_per_launch" in tj[k] and "write_KB_per_launch" in tj[k] for k in fams)
