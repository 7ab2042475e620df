#!/usr/bin/env python3
"""Regenerates tests/golden/*.json.

kat_sizes.json      the 36 compressed sizes the reference publishes in benchmarks.md
                    (lines 18,23,28 sparse 3500x3500; 63..223 corpus) -- typed in from
                    that file, not computed.
oracle_digests.json sha256 of the oracle's output (oracle/zs_oracle.c, which reproduces
                    all 36 sizes) for every corpus file at several levels, plus the
                    digests the survey's independent model published.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding  # noqa: E402

KAT = {  # file: (L1, L3, L6)   benchmarks.md
    "alice29.txt": (63340, 60207, 55818), "asyoulik.txt": (55139, 52914, 50068), "cp.html": (8907, 8645, 8233),
    "fields.c": (3766, 3570, 3280), "grammar.lsp": (1334, 1316, 1251), "kennedy.xls": (199356, 203717, 187289),
    "lcet10.txt": (167403, 160050, 147916), "plrabn12.txt": (220181, 209933, 199026), "ptt5": (67013, 60164, 59946),
    "sum": (14694, 14383, 14002), "xargs.1": (1901, 1876, 1828), "sparse3500": (825050, 825050, 659280),
}
SURVEY_DIGESTS = {  # SURVEY.md A.9 (independent C model of the survey session)
    "alice29.txt-crlf:1": "9aa96c14d388de0cafbdbfa98980d673a26515f246842147f4f8dba4b10de8eb",
    "alice29.txt-crlf:3": "0238cdad1383eececdf701609d687d3531d62cecf193e2603eef29ca7fdf6021",
    "alice29.txt-crlf:6": "6a4cf5a888111c0166050fa58b71e707139ebc9e376dddaec00bcbbf91b57f18",
    "sparse3500:1": "c3a5eb58599f82b77d507e4a0ba2bd06c19c1ba77c8e90013a8948b096ff1963",
    "sparse3500:6": "2c5321875f54e84b9af73f14bf893002aa4098ad6111611eb87d20ef4d2c14d1",
}

if __name__ == "__main__":
    json.dump({k: list(v) for k, v in KAT.items()}, open(os.path.join(HERE, "kat_sizes.json"), "w"), indent=1)
    o = oracle_binding.Oracle()
    dig = {}
    for f in sorted(os.listdir(oracle_binding.CORPUS)):
        d = oracle_binding.corpus(f)
        for lvl in (1, 2, 3, 4, 5, 6, 7, 8, 9):
            z = o.compress(d, lvl)
            dig["%s:%d" % (f, lvl)] = [len(z), hashlib.sha256(z).hexdigest()]
    json.dump({"oracle": dig, "survey_model": SURVEY_DIGESTS}, open(os.path.join(HERE, "oracle_digests.json"), "w"), indent=1)
    print("wrote", len(dig), "digests")
