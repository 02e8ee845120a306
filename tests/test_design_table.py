"""DESIGN.md's kernel table is generated from profiles/ (tools/kernel_table.py): the block in the file must be current."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_design_kernel_table_is_what_the_script_generates():
    import kernel_table as kt
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    i, j = s.index(kt.BEGIN), s.index(kt.END)
    block = s[i + len(kt.BEGIN):j].strip()
    want = kt.generate(kt.newest_round()).strip()
    assert block == want, "run: python tools/kernel_table.py --update-design"


def test_every_profile_the_table_cites_exists():
    import re
    s = open(os.path.join(ROOT, "DESIGN.md")).read()
    cited = set(re.findall(r"`(profiles/[A-Za-z0-9_./-]+)`", s))
    missing = [p for p in sorted(cited) if "*" not in p and not os.path.exists(os.path.join(ROOT, p))]
    assert not missing, missing
