"""The documents quote measured figures; the measurements live under profiles/.  Two rounds in a row a
figure in the text survived its file's replacement.  This test ties them together:

  * every profiles/rNN_* file named in DESIGN.md, LABNOTES.md (rounds 1-4: DESIGN.md as it stood then), README.md,
    INTEGRATION.md or profiles/README.md exists;
  * profiles/quoted_figures.json lists the figures the documents quote from JSON profiles -- document,
    file, path to the value, how it is formatted, and the phrase it appears in -- and every phrase,
    re-built from the FILE's value, must be in the document.  Editing a profile without its text, or a
    text without its profile, fails here.  (New quoted figures get an entry; the bench lines of the
    current round are all listed.)"""
import json
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "LABNOTES.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]
MANIFEST = os.path.join(REPO, "profiles", "quoted_figures.json")


def _text(doc):
    with open(os.path.join(REPO, doc), encoding="utf-8") as fh:
        return fh.read()


@pytest.mark.parametrize("doc", DOCS)
def test_every_profile_a_document_names_exists(doc):
    names = set(re.findall(r"(?:profiles/)?(r0\d_[A-Za-z0-9_./*{}…-]+?\.(?:jsonl|json|txt|csv))", _text(doc)))
    missing = [n for n in sorted(names) if not any(c in n for c in "*{}…")
               and not os.path.exists(os.path.join(REPO, "profiles", n))]
    assert not missing, f"{doc} names profiles that do not exist: {missing}"


def _value(entry):
    path = os.path.join(REPO, entry["file"])
    with open(path, encoding="utf-8") as fh:
        if path.endswith(".jsonl"):
            rows = [json.loads(ln) for ln in fh if ln.strip().startswith("{")]
            where = entry.get("where")
            if where:                                   # the first line whose fields match
                rows = [r for r in rows if all(r.get(k) == v for k, v in where.items())]
            obj = rows[entry.get("line", 0)]
        else:
            obj = json.load(fh)
    for key in entry["path"]:
        obj = obj[key]
    return float(obj) * float(entry.get("scale", 1.0))


def _entries():
    with open(MANIFEST, encoding="utf-8") as fh:
        return json.load(fh)


def test_manifest_is_well_formed():
    entries = _entries()
    assert len(entries) >= 8
    for e in entries:
        assert e["doc"] in DOCS and os.path.exists(os.path.join(REPO, e["file"])), e
        assert "{}" in e["quote"], e


@pytest.mark.parametrize("k", range(len(_entries()) if os.path.exists(MANIFEST) else 0))
def test_quoted_figure_matches_its_profile(k):
    e = _entries()[k]
    shown = e["fmt"].format(_value(e))
    if e.get("thin_space"):                             # 14 681: thousands separated by a blank
        shown = shown.replace(",", " ")
    phrase = e["quote"].replace("{}", shown)
    assert phrase in _text(e["doc"]), (f"{e['doc']} should say {phrase!r} -- the value in {e['file']} at "
                                       f"{e['path']} is {shown}")
