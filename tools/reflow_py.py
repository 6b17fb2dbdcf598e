#!/usr/bin/env python3
"""Brings over-long lines of the Python sources under a column limit without changing the program: inside brackets a line is broken after
a comma (or before a binary `+`, `and`, `or`, `if`, `else`, `for`) of the shallowest bracket depth, a trailing comment moves to its own line
above, a comment that is too long by itself is wrapped.  Lines inside multi-line strings are left alone.  The abstract syntax tree of the
result is compared with the original's: a file is only rewritten when they are equal.
usage: reflow_py.py [--limit 160] [--check] FILE...   (--check: only report the lines over the limit, exit 1 if any)"""
import ast
import io
import sys
import tokenize

OPEN, CLOSE = "([{", ")]}"
BEFORE = {"+", "and", "or", "if", "else", "for"}


def wrap_comment(indent, text, limit):
    words = text.lstrip("#").split()
    out, cur = [], indent + "#"
    for w in words:
        if len(cur) + 1 + len(w) > limit and cur.strip() != "#":
            out.append(cur); cur = indent + "#"
        cur += " " + w
    out.append(cur)
    return out


def reflow_source(src, limit):
    lines = src.split("\n")
    toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
    in_string = set()                                  # physical lines covered by a multi-line string token (except its first line's prefix)
    cands = {}                                         # line -> [(depth, column, kind)]: break AFTER column (kind 'a') or BEFORE it ('b')
    comment_at = {}                                    # line -> column of a trailing comment
    first_tok_col = {}
    depth = 0
    depth_at_line_start = {}
    for t in toks:
        (sr, sc), (er, ec) = t.start, t.end
        if sr not in depth_at_line_start: depth_at_line_start[sr] = depth
        if t.type == tokenize.STRING and er > sr:
            for r in range(sr, er + 1): in_string.add(r)
        if t.type == tokenize.COMMENT:
            comment_at[sr] = sc
            continue
        if t.type in (tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENDMARKER): continue
        first_tok_col.setdefault(sr, sc)
        if t.type == tokenize.OP and t.string in OPEN:
            depth += 1
            cands.setdefault(sr, []).append((depth, ec, "a"))          # right behind an opening bracket (its contents are one level deeper)
        elif t.type == tokenize.OP and t.string in CLOSE: depth -= 1
        elif depth > 0 and t.type == tokenize.OP and t.string == ",": cands.setdefault(sr, []).append((depth, ec, "a"))
        elif depth > 0 and t.type == tokenize.STRING and er == sr and t.string[0] in "\"'" and not t.string.startswith(t.string[0] * 3):
            # a plain one-line literal inside brackets may be cut into adjacent literals behind a space ("ab cd" -> "ab " "cd": the same constant)
            for k in range(2, len(t.string) - 2):
                if t.string[k] == " " and t.string[k - 1] != "\\": cands.setdefault(sr, []).append((depth + 1, sc + k + 1, t.string[0]))
        elif depth > 0 and ((t.type == tokenize.OP and t.string == "+") or (t.type == tokenize.NAME and t.string in BEFORE)) and sc > first_tok_col.get(sr, sc):
            cands.setdefault(sr, []).append((depth, sc, "b"))
    out = []
    for no, line in enumerate(lines, 1):
        if len(line) <= limit or no in in_string:
            out.append(line); continue
        stripped = line.lstrip(" ")
        indent = line[:len(line) - len(stripped)]
        if stripped.startswith("#"):
            out += wrap_comment(indent, stripped, limit); continue
        code, comment = line, None
        if no in comment_at:
            code, comment = line[:comment_at[no]].rstrip(), line[comment_at[no]:]
        if comment is not None and code.strip():
            out += wrap_comment(indent, comment, limit)
        if len(code) <= limit:
            out.append(code); continue
        pts = sorted(cands.get(no, []), key=lambda c: c[1])
        pts = [(d, c, k) for d, c, k in pts if 0 < c < len(code)]
        done = False
        for level in sorted(set(d for d, _, _ in pts)):
            cols = [(c, k) for d, c, k in pts if d <= level]
            pieces, start, cont, ok, reopen = [], 0, "", True, ""
            while True:
                room = limit - len(cont) - len(reopen)
                rest = code[start:] if start == 0 or reopen else code[start:].lstrip(" ")
                if len(rest) <= room:
                    pieces.append(cont + reopen + rest); break
                best = None
                for c, k in cols:
                    if c <= start: continue
                    seg = code[start:c] if start == 0 or reopen else code[start:c].lstrip(" ")
                    if len(seg.rstrip() if k in "ab" else seg) + (0 if k in "ab" else 1) <= room: best = (c, k)
                    else: break
                if best is None: ok = False; break
                c, k = best
                seg = code[start:c] if start == 0 or reopen else code[start:c].lstrip(" ")
                pieces.append(cont + reopen + (seg.rstrip() if k in "ab" else seg + k)); start = c
                reopen = "" if k in "ab" else k
                cont = indent + "    " + ("    " if depth_at_line_start.get(no, 0) == 0
                    and code.lstrip().startswith(("if ", "for ", "while ", "with ", "def ", "elif ", "class ")) else "")
            if ok:
                out += pieces; done = True; break
        if not done: out.append(code)
    return "\n".join(out)


def main():
    args = sys.argv[1:]
    limit, check = 160, False
    if "--limit" in args:
        k = args.index("--limit"); limit = int(args[k + 1]); del args[k:k + 2]
    if "--check" in args:
        check = True; args.remove("--check")
    bad = 0
    for path in args:
        src = open(path).read()
        if check:
            for n, l in enumerate(src.split("\n"), 1):
                if len(l) > limit: print("%s:%d: %d characters" % (path, n, len(l))); bad += 1
            continue
        new = reflow_source(src, limit)
        if new == src: continue
        if ast.dump(ast.parse(new)) != ast.dump(ast.parse(src)):
            print("%s: NOT rewritten (the syntax tree would change)" % path); bad += 1; continue
        open(path, "w").write(new)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
