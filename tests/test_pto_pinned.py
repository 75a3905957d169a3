"""The product's PTO reader (include/eu_frontend.hpp: pto_script) against the REFERENCE's own parser
(pto.h:72-180, pto_parser_type), line group by line group and field by field.

* live (only where /root/reference exists): pto.h compiled in place into oracle/_ref/libref_zimt.so
  (ref_pto_parse), both parsers on every script of tests/pto_cases.py;
* fixture: tests/golden/pto_golden.json holds the reference parser's output for the same scripts (generated
  by tests/golden/make_pto_golden.py), so the comparison also runs where the reference is absent."""
import ctypes as C
import json
import os
import subprocess

import pytest

import pto_cases
import refz

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "envutil_amd", "build", "pto_dump")
GOLDEN = os.path.join(ROOT, "tests", "golden", "pto_golden.json")


@pytest.fixture(scope="module")
def dump():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tests", "csrc", "pto_dump.cc")
    hdr = os.path.join(ROOT, "include", "eu_frontend.hpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src, "-o", EXE])

    def run(text):
        return subprocess.run([EXE], input=text.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    return run


def ref_parse(text):
    f = refz.lib().ref_pto_parse
    f.restype = C.c_long
    f.argtypes = [C.c_char_p, C.c_char_p, C.c_long]
    need = f(text.encode(), None, 0)
    buf = C.create_string_buffer(need)
    f(text.encode(), buf, need)
    return buf.value.decode()


def as_groups(dumped):
    groups = {}
    for line in dumped.splitlines():
        head, idx, *fields = line.split("\t")
        groups.setdefault(head, {})[int(idx)] = dict(f.split("=", 1) for f in fields)
    return groups


@pytest.mark.parametrize("name", sorted(pto_cases.CASES))
def test_fixture(dump, name):
    golden = json.load(open(GOLDEN))
    mine, theirs = as_groups(dump(pto_cases.CASES[name])), as_groups(golden[name])
    assert mine.keys() == theirs.keys()
    for head in theirs:
        assert mine[head] == theirs[head], head


@pytest.mark.skipif(not refz.available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", sorted(pto_cases.CASES))
def test_live(dump, name):
    text = pto_cases.CASES[name]
    assert dump(text) == ref_parse(text)
    # and the fixture is what the reference says today
    assert json.load(open(GOLDEN))[name] == ref_parse(text)
