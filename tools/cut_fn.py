#!/usr/bin/env python3
"""Developer helper: delete top-level functions / kernels (and the comment block glued above them) from a source file.
    python tools/cut_fn.py FILE name1 name2 ..."""
import re
import sys


def cut(lines, name):
    pat = re.compile(r"\b%s\b\s*(<[^;{]*>)?\s*\(" % re.escape(name))
    i = 0
    while i < len(lines):
        ln = lines[i]
        if pat.search(ln) and not ln.startswith((" ", "\t", "//")) and not ln.rstrip().endswith(";"):
            # walk to the opening brace, then to its match at column 0
            j = i
            while "{" not in lines[j]:
                if lines[j].rstrip().endswith(";"):
                    break
                j += 1
            if "{" not in lines[j]:
                i += 1
                continue
            depth, k = 0, j
            while True:
                depth += lines[k].count("{") - lines[k].count("}")
                if depth == 0:
                    break
                k += 1
            s = i
            while s > 0 and (lines[s - 1].startswith("template") or lines[s - 1].startswith("//") or
                             lines[s - 1].startswith("__launch_bounds__")):
                s -= 1
            del lines[s:k + 1]
            while s < len(lines) and s > 0 and lines[s].strip() == "" and lines[s - 1].strip() == "":
                del lines[s]
            return True
        i += 1
    return False


def main():
    path, names = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    for n in names:
        if not cut(lines, n):
            print("not found:", n)
    open(path, "w").write("\n".join(lines))


if __name__ == "__main__":
    main()
