#!/usr/bin/env python3
"""Resolves the measurement-only preprocessor branches of a HIP source the way a product build sees them and writes the result
back: `#if` / `#ifdef` / `#ifndef` / `#elif` whose condition only involves the names below are evaluated (names in UNDEF are
not defined, names in DEFINED have the given value), the dead branches and the directives themselves are dropped, everything
else is kept byte for byte.  Used once in round 6 to move the lab out of the shipped translation units: the unstripped files
live on as composer_amd/csrc/experiments/*_lab.hip (compiled only by tools/ab_build.py with -DCOMPOSER_EXPERIMENTS).
    python tools/strip_lab.py composer_amd/csrc/attention.hip [...]
"""
import re
import sys

UNDEF = {"COMPOSER_EXPERIMENTS", "ATTN_NO_HEAD_ROTATE", "ATTN_LIGHT_FIRST", "ATTN_COND_STORE", "ATTN_RESCALE_LOG2", "DEC_NO_CW2",
         "DEC_HALF_W", "DEC_ALIAS_L0", "GEMM_EARLY_SLAB", "COMPOSER_TIED_WGRAD_OLD", "COMPOSER_WGRAD_UNGROUPED"}
DEFINED = {"ATTN_DIAG": 0, "GEMM_DIAG": 0, "P4_DIAG": 0}
KNOWN = UNDEF | set(DEFINED)


def evaluate(kind, expr):
    """True / False when the condition is decided by KNOWN names alone, None otherwise."""
    expr = re.sub(r"//.*$", "", expr).strip()
    expr = re.sub(r"/\*.*?\*/", "", expr).strip()
    if kind in ("ifdef", "ifndef"):
        name = expr.split()[0]
        if name in UNDEF:
            return kind == "ifndef"
        if name in DEFINED:
            return kind == "ifdef"
        return None
    names = set(re.findall(r"[A-Za-z_]\w*", expr)) - {"defined"}
    if not names or not names <= KNOWN:
        return None
    e = re.sub(r"defined\s*\(\s*(\w+)\s*\)", lambda m: "1" if m.group(1) in DEFINED else "0", expr)
    e = re.sub(r"defined\s+(\w+)", lambda m: "1" if m.group(1) in DEFINED else "0", e)
    e = re.sub(r"[A-Za-z_]\w*", lambda m: str(DEFINED.get(m.group(0), 0)), e)
    e = e.replace("&&", " and ").replace("||", " or ")
    e = re.sub(r"!(?!=)", " not ", e)
    return bool(eval(e, {"__builtins__": {}}))


def strip(text):
    out = []
    # stack entries: [resolved (bool), emitting_parent, taken_already, currently_true]
    stack = []
    emitting = True
    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)$", line)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            v = evaluate(kind, rest) if emitting else False
            if not emitting:
                stack.append([True, False, True, False])
            elif v is None:
                stack.append([False, True, False, True])
                out.append(line)
            else:
                stack.append([True, True, v, v])
                emitting = v
        elif kind == "elif":
            top = stack[-1]
            if not top[0]:
                out.append(line)
            elif top[1]:
                if top[2]:
                    emitting = False
                    top[3] = False
                else:
                    v = evaluate("if", rest)
                    if v is None:
                        raise SystemExit("strip_lab: '#elif %s' after a resolved '#if' cannot be resolved" % rest.strip())
                    top[2] = top[3] = v
                    emitting = v
        elif kind == "else":
            top = stack[-1]
            if not top[0]:
                out.append(line)
            elif top[1]:
                emitting = not top[2]
                top[2] = True
        else:
            top = stack.pop()
            if not top[0]:
                out.append(line)
            else:
                emitting = top[1]
    assert not stack
    text = "\n".join(out)
    # the "#undef X / #define X 0" pair left behind by a resolved "ladders exist in experiments builds only" guard
    for name in DEFINED:
        text = re.sub(r"#undef %s\n#define %s 0\n" % (name, name), "", text)
    return text


if __name__ == "__main__":
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        open(path, "w").write(new)
        print("%s: %d -> %d lines" % (path, src.count("\n"), new.count("\n")))
