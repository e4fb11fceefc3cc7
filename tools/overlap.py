#!/usr/bin/env python3
"""Normalised line overlap of a repo file with a reference file: comments and whitespace stripped, lines of fewer than
8 characters (braces, else, ...) ignored; prints the share of the repo file's significant lines found verbatim in the
reference file.  Usage: tools/overlap.py <repo file> <reference file> [...more reference files]"""
import re
import sys


def significant(path):
    text = open(path, errors="replace").read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = []
    for line in text.split("\n"):
        line = re.sub(r"//.*", "", line)
        line = re.sub(r"\s+", "", line)
        if len(line) >= 8:
            out.append(line)
    return out


def main():
    mine = significant(sys.argv[1])
    theirs = set()
    for ref in sys.argv[2:]:
        theirs.update(significant(ref))
    hit = [l for l in mine if l in theirs]
    print(f"{sys.argv[1]}: {len(hit)}/{len(mine)} = {100.0 * len(hit) / max(1, len(mine)):.1f} %")
    if "-v" in sys.argv:
        for l in hit:
            print("   ", l)


if __name__ == "__main__":
    main()
