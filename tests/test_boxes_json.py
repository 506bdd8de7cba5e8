"""bboxes_3D_cam0/BBoxes_<frame>.json read by the library (lpf_parse_boxes_json; the read-ahead reader's worker uses the same parser)
against json.load, which is what the reference uses (load_bounding_boxes, V3:31-38; cvs_erosion.py:333): the same indices and,
bit for bit, the same doubles -- or the file is reported as not interpreted and json.load's result / error stands.  No GPU involved."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from lidar_object_detection_amd import _native


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _write(tmp_path, text, name="BBoxes_1.json"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def _box_text(numbers, index="7"):
    assert len(numbers) == 24
    rows = ", ".join("[" + ", ".join(numbers[3 * r:3 * r + 3]) + "]" for r in range(8))
    return '{"index": %s, "corners_cam0": [%s]}' % (index, rows)


def test_golden_frames_box_files_parse_like_json_load(tmp_path):
    for name in ("frame_0000000100.npz", "frame_0000002449_full.npz", "frame_0000001461_full.npz"):
        g = np.load(os.path.join(GOLDEN, name))
        raw = [{"index": int(1000 + 3 * i), "corners_cam0": c.tolist()} for i, c in enumerate(g["corners_cam0_raw"])]
        for indent in (None, 2):
            p = _write(tmp_path, json.dumps(raw, indent=indent))
            st, idx, cs = _native.parse_boxes_file(p)
            want = json.load(open(p))
            assert st == _native.BOXES_PARSED and idx.dtype == np.int32 and cs.dtype == np.float64 and cs.shape == (len(want), 8, 3)
            assert idx.tolist() == [b["index"] for b in want]
            assert np.array_equal(_bits(cs), _bits(np.array([b["corners_cam0"] for b in want])))
            assert np.array_equal(_bits(cs), _bits(g["corners_cam0_raw"]))                # (and json round-trips a double)


HARD = ["2.2250738585072011e-308", "2.2250738585072012e-308", "2.2250738585072014e-308", "4.9e-324", "5e-324",
        "2.4703282292062327e-324", "2.4703282292062328e-324", "1.7976931348623157e308", "1.7976931348623158e308",
        "1.7976931348623159e308", "1e309", "-1e400", "9007199254740993", "9007199254740992", "9007199254740995", "0.1", "1e23", "8.41e21",
        "9.5367431640625e-07", "-0.0", "-0", "0", "0e0", "1E+2", "1e-2", "123456789012345678901234567890123456789012345678901234567890.5e-40",
        "0." + "0" * 70 + "1234567890123456789", "6.02214076e23", "1e-400", "-108.11953572416087", "3.141592653589793238462643383279502884",
        "0.30000000000000004", "1.0000000000000002", "1.00000000000000011102230246251565404236316680908203125",
        "1.00000000000000011102230246251565404236316680908203126", "1.00000000000000011102230246251565404236316680908203124",
        "17976931348623157" + "0" * 292, "4e-324", "2e-324", "3e-324"]


def test_hard_numbers_round_as_python_rounds_them(tmp_path):
    toks = HARD + ["1.5"] * (-len(HARD) % 24)
    boxes = [_box_text(toks[i:i + 24], index=str(i)) for i in range(0, len(toks), 24)]
    p = _write(tmp_path, "[" + ",\n".join(boxes) + "]\n")
    st, idx, cs = _native.parse_boxes_file(p)
    want = json.load(open(p))
    assert st == _native.BOXES_PARSED and idx.tolist() == [b["index"] for b in want]
    w = np.array([b["corners_cam0"] for b in want], dtype=np.float64)
    assert np.array_equal(_bits(cs), _bits(w)), [(t, a, b) for t, a, b in zip(toks, cs.ravel().tolist(), w.ravel().tolist()) if a != b or np.signbit(a) != np.signbit(b)]
    # ("-0.0" is a float, -0.0; "-0" is an int to json.load, and the float64 array holds +0.0 for it)
    assert np.isinf(cs).sum() >= 3 and np.signbit(cs.ravel()[toks.index("-0.0")]) and not np.signbit(cs.ravel()[toks.index("-0")])


def test_random_doubles_by_their_repr(tmp_path):
    rng = np.random.default_rng(11)
    v = rng.integers(0, 2 ** 64, 24 * 400, dtype=np.uint64).view(np.float64)
    v = np.where(np.isfinite(v), v, 1.0).reshape(-1, 8, 3)
    raw = [{"corners_cam0": c.tolist(), "index": int(i) - 200} for i, c in enumerate(v)]          # (keys in the other order, negative indices)
    p = _write(tmp_path, json.dumps(raw))
    st, idx, cs = _native.parse_boxes_file(p)
    assert st == _native.BOXES_PARSED and np.array_equal(_bits(cs), _bits(v)) and idx.tolist() == list(range(-200, 200))


def test_whitespace_empty_lists_and_index_range(tmp_path):
    one = _box_text(["1", "2.5", "-3e0"] * 8, index="-2147483648")
    st, idx, cs = _native.parse_boxes_file(_write(tmp_path, " \n\t[ \r\n" + one.replace(",", " ,\n ").replace(":", " :\t") + " ]  \n"))
    assert st == _native.BOXES_PARSED and idx.tolist() == [-2147483648] and cs[0, 0].tolist() == [1.0, 2.5, -3.0]
    for text in ("[]", " [ ] \n"):
        st, idx, cs = _native.parse_boxes_file(_write(tmp_path, text))
        assert st == _native.BOXES_PARSED and len(idx) == 0 and cs.shape == (0, 8, 3)
    st, idx, _ = _native.parse_boxes_file(_write(tmp_path, "[" + _box_text(["0"] * 24, index="2147483647") + "]"))
    assert st == _native.BOXES_PARSED and idx.tolist() == [2147483647]


GOOD = _box_text(["1", "2", "3"] * 8)
OTHER = {
    "extra key": '[{"index": 1, "corners_cam0": %s, "label": "car"}]' % GOOD[GOOD.index("[["):-1],
    "float index": "[" + _box_text(["1"] * 24, index="1.0") + "]",
    "exponent index": "[" + _box_text(["1"] * 24, index="1e2") + "]",
    "index beyond int32": "[" + _box_text(["1"] * 24, index="2147483648") + "]",
    "string index": "[" + _box_text(["1"] * 24, index='"3"') + "]",
    "NaN": "[" + _box_text(["NaN"] + ["1"] * 23) + "]",
    "Infinity": "[" + _box_text(["-Infinity"] + ["1"] * 23) + "]",
    "trailing text": "[" + GOOD + "] x",
    "two documents": "[" + GOOD + "][]",
    "trailing comma": "[" + GOOD + ",]",
    "missing bracket": "[" + GOOD,
    "seven corners": '[{"index": 1, "corners_cam0": [%s]}]' % ", ".join(["[1, 2, 3]"] * 7),
    "nine corners": '[{"index": 1, "corners_cam0": [%s]}]' % ", ".join(["[1, 2, 3]"] * 9),
    "four coordinates": '[{"index": 1, "corners_cam0": [%s]}]' % ", ".join(["[1, 2, 3, 4]"] * 8),
    "a dict": "{" + GOOD[1:],
    "only an index": '[{"index": 1}]',
    "duplicate key": '[{"index": 1, "index": 2, "corners_cam0": %s]' % GOOD[GOOD.index("[["):],
    "escaped key": "[" + GOOD.replace('"index"', '"inde\\u0078"') + "]",
    "plus sign": "[" + _box_text(["+1"] + ["1"] * 23) + "]",
    "leading zero": "[" + _box_text(["01"] + ["1"] * 23) + "]",
    "bare fraction": "[" + _box_text([".5"] + ["1"] * 23) + "]",
    "dangling point": "[" + _box_text(["1."] + ["1"] * 23) + "]",
    "hex": "[" + _box_text(["0x10"] + ["1"] * 23) + "]",
    "an integer beyond the doubles (np.array raises OverflowError)": "[" + _box_text(["1" + "0" * 400] + ["1"] * 23) + "]",
    "null corner": "[" + _box_text(["null"] + ["1"] * 23) + "]",
    "empty file": "",
    "blank file": "  \n",
    "a list of numbers": "[1, 2, 3]",
    "byte order mark": "﻿[" + GOOD + "]",
}


@pytest.mark.parametrize("case", sorted(OTHER))
def test_anything_else_is_left_to_the_json_library(tmp_path, case):
    p = _write(tmp_path, OTHER[case])
    st, idx, cs = _native.parse_boxes_file(p)
    assert st == _native.BOXES_OTHER and len(idx) == 0 and cs.shape == (0, 8, 3), case
    # ... because json.load either refuses the file or gives something that is not the plain schema
    try:
        got = json.load(open(p))
    except ValueError:
        return
    def finite_array(c):
        try:
            return bool(np.isfinite(np.array(c, np.float64)).all())
        except OverflowError:
            return False
    plain = isinstance(got, list) and all(isinstance(b, dict) and set(b) == {"index", "corners_cam0"} and type(b["index"]) is int and
                                          -2 ** 31 <= b["index"] < 2 ** 31 and np.shape(b["corners_cam0"]) == (8, 3) and
                                          finite_array(b["corners_cam0"]) for b in got)
    assert not plain or case in ("duplicate key", "escaped key"), case      # (json.load reads these two: the parser just does not go there)


def test_every_prefix_of_a_file_is_refused_not_misread(tmp_path):
    text = "[" + _box_text(["-1.25e1", "2", "3"] * 8, index="12") + ", " + _box_text(["4"] * 24, index="-5") + "]"
    for k in range(len(text)):
        st, idx, _ = _native.parse_boxes_file(_write(tmp_path, text[:k]))
        assert st == _native.BOXES_OTHER, (k, text[:k])
    st, idx, _ = _native.parse_boxes_file(_write(tmp_path, text))
    assert st == _native.BOXES_PARSED and idx.tolist() == [12, -5]


def test_absent_file_and_capacity(tmp_path):
    st, idx, cs = _native.parse_boxes_file(str(tmp_path / "BBoxes_404.json"))
    assert st == _native.BOXES_ABSENT and len(idx) == 0
    st, _, _ = _native.parse_boxes_file(str(tmp_path))                       # a directory: not interpreted (open() raises for the caller)
    assert st == _native.BOXES_OTHER
    lib = _native.load()
    p = _write(tmp_path, "[" + ", ".join([GOOD] * 3) + "]")
    cs, ix = np.zeros((2, 8, 3)), np.zeros(2, np.int32)
    n, state = ctypes.c_int(0), ctypes.c_int(-1)
    rc = lib.lpf_parse_boxes_json(p.encode(), cs.ctypes.data, ix.ctypes.data, 2, ctypes.byref(n), ctypes.byref(state))
    assert rc == -1 and n.value == 3 and state.value == _native.BOXES_PARSED and not cs.any()      # LPF_ERR_ARG, nothing written
    assert lib.lpf_parse_boxes_json(None, None, None, 0, ctypes.byref(n), ctypes.byref(state)) == -1


def test_integration_md_box_file_snippet_runs(tmp_path, capsys):
    """The box-file snippet of INTEGRATION.md section C, executed as written through raw ctypes: a plain file, a file with another key
    (json.load's result), an absent file (the reference's message) and an empty list."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = text[text.index("## C. Raw ctypes stub"):]
    code = [b for b in re.findall(r"```python\n(.*?)```", sec, re.S) if "lpf_parse_boxes_json" in b][0]
    lib = ctypes.CDLL(_native.library_path())
    ns = {"ctypes": ctypes, "np": np, "os": os, "json": json, "lib": lib, "P": ctypes.c_void_p}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    load = ns["load_bounding_boxes_cam0"]
    g = np.load(os.path.join(GOLDEN, "frame_0000000100.npz"))
    raw = [{"index": int(i), "corners_cam0": c.tolist()} for i, c in zip(g["box_index_raw"], g["corners_cam0_raw"])]
    idx, cs = load(_write(tmp_path, json.dumps(raw)))
    assert idx.tolist() == g["box_index_raw"].tolist() and np.array_equal(_bits(cs), _bits(g["corners_cam0_raw"]))
    for b in raw:
        b["label"] = "car"
    idx2, cs2 = load(_write(tmp_path, json.dumps(raw), "BBoxes_2.json"))
    assert idx2.tolist() == idx.tolist() and np.array_equal(_bits(cs2), _bits(cs))
    assert load(_write(tmp_path, "[]", "BBoxes_3.json")) is None
    capsys.readouterr()
    assert load(str(tmp_path / "BBoxes_4.json")) is None and "No bounding boxes found" in capsys.readouterr().out
