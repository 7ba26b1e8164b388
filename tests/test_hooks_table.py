"""CPU: the library's test hooks / tuning switches live in ONE table (hylight_amd/csrc/runtime.cpp: HOOK_NAMES, read from the
environment once per C-ABI call by hooks_refresh) - no other getenv in the library - and DESIGN.md section 8 lists every one
of them."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "hylight_amd", "csrc")


def _table():
    txt = open(os.path.join(CSRC, "runtime.cpp")).read()
    body = txt[txt.index("HOOK_NAMES[] = {"):]
    body = body[:body.index("};")]
    return sorted(set(re.findall(r'"(HLMI_[A-Z0-9_]+)"', body)))


def test_no_getenv_outside_the_hook_table():
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".cpp", ".h")):
            continue
        txt = open(os.path.join(CSRC, f)).read()
        n = len(re.findall(r"\bgetenv\s*\(", txt))
        assert n == (1 if f == "runtime.cpp" else 0), f"{f}: {n} getenv call(s)"


def test_every_hook_used_is_in_the_table_and_documented():
    names = _table()
    assert len(names) >= 25
    used = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".cpp", ".h")):
            used |= set(re.findall(r'hook\("(HLMI_[A-Z0-9_]+)"\)', open(os.path.join(CSRC, f)).read()))
    assert used <= set(names), sorted(used - set(names))
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    section = design[design.index("## 8. Environment switches"):design.index("## 9.")]
    missing = [n for n in names if n not in section]
    assert not missing, f"not in DESIGN.md section 8: {missing}"
