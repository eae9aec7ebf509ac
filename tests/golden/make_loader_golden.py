#!/usr/bin/env python3
"""Golden fixtures for the table loader, produced by the REFERENCE's own `read_csv`.

Runs only in the build container (reference mounted at /root/reference).  Committed output:
  tests/golden/tbl/*.tbl          small dbgen-style text tables — DATA written by this repository's
                                  generator (sdqlpy_amd/tpch.py, SF=0.0002) plus two hand-made
                                  edge-case files; nothing of the reference is in them
  tests/golden/tbl_expected.npz   the arrays `sdqlpy.sdql_lib.read_csv` (reference
                                  src/sdqlpy/sdql_lib.py:69-129, imported unmodified) returns for
                                  each file, keyed "<file>/<column>"

    python tests/golden/make_loader_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_ROOT = "/root/reference"

from sdqlpy_amd import tpch  # noqa: E402
from sdqlpy_amd import sdql_lib as mine  # noqa: E402

TBL = os.path.join(HERE, "tbl")

# hand-made files: every line has the 5 columns (int, float, date, string(6), string(1) "NA")
EDGE_SCHEMA = [("e_key", int), ("e_val", float), ("e_day", "date"), ("e_txt", ("string", 6)), ("e_NA", ("string", 1))]
EDGE_PLAIN = [                       # the native parser takes all of these
    "1|0.00|1992-01-01|abc|",
    "-7|-0.5|1998-12-31|abcdefXYZ|",             # text longer than its width is cut to 6 units
    "+5|.25|2000-02-29||",                       # leading '+', bare fraction, empty text
    "9223372036854775807|1e3|0001-01-01|né ü|",   # INT64_MAX, exponent, non-ASCII text
    "-9223372036854775808|2.5E-3|9999-12-31|日本語テキスト|",
    "0|123456789.125|1995-03-15|a b c|",
    "42|0.1|1995-3-5|tab\there|",                # date without zero padding: digits are just joined
]
EDGE_GENERAL = [                     # these need the csv-module path (quotes, spaces, underscores, blank line)
    '1|1.5|1992-01-01|"a|b"|',
    ' 8 | 2.5 |1992-01-02|sp|',
    '1_000|1_0.5|1992-01-03|"q""t"|',
    '',
    '3|inf|1992-01-04|x|',
    '4|nan|1992-01-05|y|',
]


def ref_schema(ref, fields):
    """{record({...}): bool} built from the REFERENCE's own type markers."""
    d = {}
    for name, t in fields:
        if t is int or t is float:
            d[name] = t
        elif t == "date":
            d[name] = ref.date
        else:
            d[name] = ref.string(t[1])
    return {ref.record(d): bool}


def fields_of(table):
    out = []
    for name, t in next(iter(tpch.SCHEMAS[table].keys())).getContainer().items():
        if t is int or t is float:
            out.append((name, t))
        elif t is mine.date:
            out.append((name, "date"))
        else:
            out.append((name, ("string", t.max_size)))
    return out


def main():
    sys.path.insert(0, os.path.join(REF_ROOT, "src"))
    import sdqlpy.sdql_lib as ref
    os.makedirs(TBL, exist_ok=True)
    db = tpch.generate(0.0002, tpch.DEFAULT_SEED, tables=sorted(tpch.SCHEMAS))
    paths = tpch.write_tbl(TBL, db)
    files = {os.path.basename(p): fields_of(t) for t, p in paths.items()}
    for name, lines in (("edge_plain.tbl", EDGE_PLAIN), ("edge_general.tbl", EDGE_GENERAL)):
        with open(os.path.join(TBL, name), "w", newline="\n") as fh:
            fh.write("\n".join(lines) + "\n")
        files[name] = EDGE_SCHEMA
    with open(os.path.join(TBL, "edge_unterminated.tbl"), "w", newline="\n") as fh:   # last line without '\n'
        fh.write("\n".join(EDGE_PLAIN[:3]))
    files["edge_unterminated.tbl"] = EDGE_SCHEMA
    open(os.path.join(TBL, "edge_empty.tbl"), "w").close()
    files["edge_empty.tbl"] = EDGE_SCHEMA
    expected = {}
    for fname, fields in sorted(files.items()):
        table = ref.read_csv(os.path.join(TBL, fname), ref_schema(ref, fields), fname)
        c = table.getContainer()
        for h, a in zip(c["headers"], c["data"]):
            expected["%s/%s" % (fname, h)] = np.asarray(a)
        print(fname, [(h, a.dtype.str, a.shape) for h, a in zip(c["headers"], c["data"])][:4], "...")
    np.savez_compressed(os.path.join(HERE, "tbl_expected.npz"), **expected)
    size = sum(os.path.getsize(os.path.join(TBL, f)) for f in os.listdir(TBL))
    print("wrote %d files (%d bytes) + tbl_expected.npz (%d bytes)" % (len(files), size, os.path.getsize(os.path.join(HERE, "tbl_expected.npz"))))


if __name__ == "__main__":
    main()
