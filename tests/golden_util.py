"""Loading tests/golden fixtures and mapping a glistcompare argv to set-operation parameters."""
from __future__ import annotations

import json
import os

import numpy as np

from genometester4_amd.listio import RECORD_DTYPE, header_bytes

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")

RULE_NAMES = {"default": 0, "add": 1, "sum": 1, "subtract": 2, "min": 3, "max": 4, "first": 5, "second": 6}
OP_FILES = {1: "union", 2: "intrsec", 4: "0_diff1", 8: "0_diff2"}

_cache = {}


def load():
    if not _cache:
        with open(os.path.join(GOLDEN, "cases.json")) as f:
            _cache["cases"] = json.load(f)
        inp = np.load(os.path.join(GOLDEN, "inputs.npz"))
        meta = json.loads(bytes(inp["__meta__"]).decode())
        _cache["inputs"] = {n: (inp[n].astype(RECORD_DTYPE), meta[n][0], meta[n][1]) for n in meta}
        _cache["outputs"] = np.load(os.path.join(GOLDEN, "outputs.npz"))
    return _cache["cases"], _cache["inputs"], _cache["outputs"]


def parse_argv(argv):
    """The subset of glistcompare's argv grammar the fixtures use (reference src/glistcompare.c:107-230)."""
    p = dict(files=[], ops=0, rule=0, cutoff=1, subtract=0, count_override=1, count_only=False, out="out",
             stream=False)
    i = 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("-"):
            p["files"].append(a)
        elif a == "-u":
            p["ops"] |= 1
        elif a == "-i":
            p["ops"] |= 2
        elif a == "-d":
            p["ops"] |= 4
        elif a == "-dd":
            p["ops"] |= 4 | 8
        elif a == "-du":
            p["ops"] |= 4
            p["subtract"] = 1
        elif a == "-c":
            i += 1
            p["cutoff"] = int(argv[i]) & 0xFFFFFFFF
        elif a == "-o":
            i += 1
            p["out"] = argv[i]
        elif a == "-r":
            i += 1
            if argv[i][0] in "123456789":
                p["rule"] = 7
                p["count_override"] = int(argv[i])
            elif argv[i] in RULE_NAMES:
                p["rule"] = RULE_NAMES[argv[i]]
        elif a == "--count_only":
            p["count_only"] = True
        elif a == "--stream":
            p["stream"] = True
        i += 1
    return p


def list_file_bytes(word_length, n_words, total_count, records):
    return header_bytes(word_length, n_words, total_count) + np.ascontiguousarray(records, dtype=RECORD_DTYPE).tobytes()


def input_records(inputs, filename):
    rec, k, _ = inputs[filename[:-5]]
    return rec, k
