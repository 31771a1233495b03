#!/usr/bin/env python3
"""Removes developer switches (#ifdef blocks of macros that the shipped build never defines) from kernel sources, keeping the shipped
branch: `python tools/strip_switches.py file ...` rewrites the files in place.  Used once per round to keep timing / debug builds out of
the shipping sources (they live in tools/exp/dev_switches.patch and are applied to a scratch copy by tools/mkvar.sh)."""
import re
import sys

PREFIXES = ("PQ_T_", "PQ_DBG", "PQ_SPAN", "PQ_NO_VOIDSKIP", "FW_DBG", "FW_T_", "FW_PLAIN_X", "GW_DBG", "WN_CLK", "GD_DBG", "RW_T_", "RW_DBG", "RW_XF_W",
            "DEC_T_", "DEC_CLK", "WN_SPLIT_PLAIN", "WN_NO_SGB", "EP_T_", "EP_DBG", "ER_T_", "ER_DBG", "GR_T_", "GR_DBG")


def is_dev(m):
    return m.startswith(PREFIXES)


def evaluate(line):
    """-> True / False when the condition only involves developer macros (all undefined), else None (keep the directive)."""
    s = line.strip()
    m = re.match(r"#\s*ifdef\s+(\w+)", s)
    if m:
        return False if is_dev(m.group(1)) else None
    m = re.match(r"#\s*ifndef\s+(\w+)", s)
    if m:
        return True if is_dev(m.group(1)) else None
    m = re.match(r"#\s*if\s+(.*?)(//.*)?$", s)
    if m:
        expr = m.group(1)
        names = re.findall(r"defined\s*\(\s*(\w+)\s*\)", expr)
        rest = re.sub(r"!?\s*defined\s*\(\s*\w+\s*\)", "", expr)
        if names and all(is_dev(n) for n in names) and re.fullmatch(r"[\s&|()]*", rest):
            py = re.sub(r"defined\s*\(\s*\w+\s*\)", "False", expr).replace("&&", " and ").replace("||", " or ").replace("!", " not ")
            return bool(eval(py))
    return None


def strip(text):
    out, stack = [], []          # stack entries: [keep_directives(bool: not ours), emitting(bool), seen_true(bool)]
    for line in text.split("\n"):
        s = line.strip()
        if re.match(r"#\s*if", s):
            v = evaluate(line)
            parent = all(e[1] for e in stack)
            if v is None:
                stack.append([True, True, True])
                if parent:
                    out.append(line)
            else:
                stack.append([False, v, v])
            continue
        if re.match(r"#\s*else", s) and stack:
            e = stack[-1]
            if e[0]:
                if all(x[1] for x in stack):
                    out.append(line)
            else:
                e[1] = not e[2]
            continue
        if re.match(r"#\s*elif", s) and stack and not stack[-1][0]:
            raise SystemExit("elif on a developer switch: not handled")
        if re.match(r"#\s*endif", s) and stack:
            e = stack.pop()
            if e[0] and all(x[1] for x in stack):
                out.append(line)
            continue
        if all(e[1] for e in stack):
            out.append(line)
    assert not stack
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        t = open(p).read()
        n = strip(t)
        if n != t:
            open(p, "w").write(n)
            print("stripped", p)
