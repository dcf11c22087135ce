#!/usr/bin/env python3
"""Linker version script for libsimdcomp_hip.so: the functions the two public headers declare are exported, every
other symbol (helpers, kernel stubs, C++ runtime details) stays local to the library.

    python3 tools/gen_exports.py include/simdcomp_hip.h include/simdcomp.h > exports.map
"""
import re
import sys


def declared_functions(path):
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"\btypedef\b[^;]*;", "", text)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n != "defined"))


def main():
    names = sorted(set(n for p in sys.argv[1:] for n in declared_functions(p)))
    if not names:
        sys.exit("no declarations found")
    print("{\n  global:")
    for n in names:
        print(f"    {n};")
    print("  local:\n    *;\n};")


if __name__ == "__main__":
    main()
