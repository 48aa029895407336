#!/usr/bin/env python3
"""Golden transcripts of the reference's READER diagnostics and debug output.

    python tests/golden/make_golden_diag.py        (build container: needs oracle/_ref/glistcompare)

Crafts malformed and unusual input files from this repo's own fixtures -- wrong tag, wrong major
version, truncated body, files shorter than a header, a version-4.4 header, malformed GT4I indices --
runs the REFERENCE glistcompare on them (also with -D and --stream) and records exit code, stdout,
stderr and created files.  Writes tests/golden/diag_cases.json and tests/golden/diag_files.npz (the
crafted input files and the reference's output files, raw bytes).  Data only.

Reference: gt4_word_map_new diagnostics src/word-map.c:181-215, gt4_index_map_new
src/index-map.c:317-373, GT4WordListStream src/word-list-stream.c:127-186, debug prints
src/glistcompare.c:224-225, :809-812, :914."""
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from genometester4_amd.listio import RECORD_DTYPE, write_list  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "glistcompare")


def main():
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    inp = np.load(os.path.join(HERE, "inputs.npz"))
    meta = json.loads(bytes(inp["__meta__"]).decode())
    idx = np.load(os.path.join(HERE, "index_inputs.npz"))
    work = tempfile.mkdtemp(prefix="gt4diag_")
    files = {}

    def put(name, data):
        files[name] = np.frombuffer(bytes(data), dtype=np.uint8)
        with open(os.path.join(work, name), "wb") as f:
            f.write(bytes(data))

    for n in ("A8", "B8", "M0", "M1", "M2"):
        write_list(os.path.join(work, n + ".list"), inp[n].astype(RECORD_DTYPE), meta[n][0])
    good = open(os.path.join(work, "A8.list"), "rb").read()
    for n in ("A8", "B8", "M0", "M1", "M2"):
        files[n + ".list"] = np.frombuffer(open(os.path.join(work, n + ".list"), "rb").read(), dtype=np.uint8)
    put("badtag.list", b"XXXX" + good[4:])
    put("major5.list", good[:4] + struct.pack("<I", 5) + good[8:])
    put("major3.list", good[:4] + struct.pack("<I", 3) + good[8:])
    put("trunc.list", good[: 48 + 12 * 5 + 7])
    put("hdr20.list", good[:20])
    put("minor4.list", good[:8] + struct.pack("<I", 4) + good[12:])
    put("minor9.list", good[:8] + struct.pack("<I", 9) + good[12:])
    # GT4I indices: any index fixture of this repo (made by the reference's glistmaker --index)
    iname = sorted(k for k in idx.files if not k.startswith("__"))[0]
    ibytes = bytes(idx[iname])
    put("good.index", ibytes)
    put("badtag.index", b"XXXX" + ibytes[4:])
    put("major5.index", ibytes[:4] + struct.pack("<I", 5) + ibytes[8:])

    cases, outputs = [], {}

    def run(cid, argv):
        before = set(os.listdir(work))
        p = subprocess.run([REF] + argv, cwd=work, capture_output=True)
        created = sorted(set(os.listdir(work)) - before)
        for f in created:
            with open(os.path.join(work, f), "rb") as fh:
                outputs["%s/%s" % (cid, f)] = np.frombuffer(fh.read(), dtype=np.uint8)
            os.remove(os.path.join(work, f))
        if p.returncode < 0:
            print("skipped (the reference died with signal %d): %s" % (-p.returncode, cid))
            return
        cases.append(dict(id=cid, tool="glistcompare", argv=argv, exit=p.returncode, stdout=p.stdout.decode("latin-1"),
                          stderr=p.stderr.decode("latin-1"), files=created))

    for bad in ("badtag.list", "major5.list", "major3.list", "trunc.list", "hdr20.list", "badtag.index", "major5.index"):
        run("diag_first_" + bad.replace(".", "_"), [bad, "B8.list", "-u", "-o", "d"])
        run("diag_second_" + bad.replace(".", "_"), ["A8.list", bad, "-i", "-o", "d"])
    run("diag_two_bad", ["badtag.list", "trunc.list", "-u"])
    run("diag_bad_in_multi", ["M0.list", "major5.list", "M1.list", "-u", "-o", "d"])
    for ok in ("minor4.list", "minor9.list"):
        run("diag_" + ok.replace(".", "_"), [ok, "B8.list", "-u", "-i", "-d", "-dd", "-o", "d"])
    # --stream: every two-file op set, cutoff and rule; N-way; with a 4.4 header
    run("stream_u_c2", ["A8.list", "B8.list", "-u", "-c", "2", "--stream", "-o", "s"])
    run("stream_i_rmax", ["A8.list", "B8.list", "-i", "-r", "max", "--stream", "-o", "s"])
    run("stream_du_dd", ["A8.list", "B8.list", "-du", "-dd", "--stream", "-o", "s"])
    run("stream_count_only", ["A8.list", "B8.list", "-u", "-i", "-d", "-dd", "--stream", "--count_only"])
    run("stream_multi_u", ["M0.list", "M1.list", "M2.list", "-u", "--stream", "-o", "s"])
    run("stream_multi_i_c2", ["M0.list", "M1.list", "M2.list", "-i", "-c", "2", "--stream", "-o", "s"])
    run("stream_minor4", ["minor4.list", "B8.list", "-u", "-i", "--stream", "-o", "s"])
    run("stream_noscouts", ["A8.list", "B8.list", "-u", "--stream", "--disable_scouts", "-o", "s"])
    # -D: the two-file path's debug lines are deterministic (the N-way path prints a rate: not pinned)
    run("debug_pair_all", ["A8.list", "B8.list", "-u", "-i", "-d", "-dd", "-D", "-o", "g"])
    run("debug_pair_u", ["A8.list", "B8.list", "-u", "-D", "-o", "g"])
    run("debug_pair_count_only", ["A8.list", "B8.list", "-u", "-i", "--count_only", "-D"])
    run("debug_pair_DD_du", ["A8.list", "B8.list", "-du", "-D", "-D", "-r", "first", "-i", "-o", "g"])
    run("debug_print_operation", ["A8.list", "B8.list", "-u", "-D", "--print_operation", "-o", "g"])

    np.savez_compressed(os.path.join(HERE, "diag_files.npz"), **{"in/" + k: v for k, v in files.items()}, **{"out/" + k: v for k, v in outputs.items()})
    with open(os.path.join(HERE, "diag_cases.json"), "w") as f:
        json.dump(cases, f, indent=0)
    shutil.rmtree(work)
    print("wrote %d cases" % len(cases))
    for c in cases:
        print(c["id"], c["exit"], repr(c["stderr"][:200]), c["files"])


if __name__ == "__main__":
    main()
