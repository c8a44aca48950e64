"""Evidence hygiene (CPU): every file under profiles/ that a committed JSON names as the source of a number — the PMC summaries behind bench.py's roofline.traffic and
valu_view, the per-round bench lines — exists and is not empty.  (Round 4 shipped a zero-byte profiles/r04_pmc_sq.txt that two JSON files cited.)"""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def _strings(obj):
    if isinstance(obj, str):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _strings(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from _strings(v)


def _cited_paths(text):
    return set(re.findall(r"profiles/[A-Za-z0-9_.\-/]+?\.(?:txt|json|csv|md)", text))


def test_every_cited_profile_exists_and_is_not_empty():
    cited = {}
    jsons = sorted(glob.glob(os.path.join(PROFILES, "*.json")))
    assert jsons, "profiles/ holds the committed measurements"
    for path in jsons:
        if re.match(r"r0[1-4]_", os.path.basename(path)):
            continue  # bench outputs of earlier rounds are kept as they were printed; what is read today (pmc_traffic*.json) and this round's outputs are held to the rule
        raw = open(path).read().strip()
        docs = []
        try:
            docs.append(json.loads(raw))
        except ValueError:
            for line in raw.splitlines():  # a bench output: one JSON object per line
                line = line.strip()
                if line.startswith("{"):
                    docs.append(json.loads(line))
        for doc in docs:
            for s in _strings(doc):
                for p in _cited_paths(s):
                    cited.setdefault(p, set()).add(os.path.basename(path))
    assert any("pmc" in p for p in cited), "the PMC summaries are cited by pmc_traffic*.json"
    missing = {p: sorted(by) for p, by in cited.items() if not os.path.isfile(os.path.join(ROOT, p)) or os.path.getsize(os.path.join(ROOT, p)) == 0}
    assert not missing, "cited but absent or empty: %r" % missing


def test_no_profile_file_is_empty():
    empty = [os.path.relpath(p, ROOT) for p in glob.glob(os.path.join(PROFILES, "*")) if os.path.isfile(p) and os.path.getsize(p) == 0]
    assert not empty, empty


def test_bench_reads_only_committed_pmc_sources():
    """bench.py prints roofline.traffic / valu_view from profiles/pmc_traffic*.json: the files it opens are there, and name the configuration they were taken on."""
    for name in ("pmc_traffic.json", "pmc_traffic_cfg4.json"):
        doc = json.load(open(os.path.join(PROFILES, name)))
        assert doc["traffic_bytes_per_launch"] > 0 and "config" in doc and "source" in doc
