#!/usr/bin/env python3
"""Brings over-long lines of the C++ host sources under a column limit by white-space changes only (there is no clang-format in the image):
a trailing comment moves to its own line above; a one-line block `head { a; b; }` is opened up; several statements on one line are put on
one line each; what is still too long is broken after a comma / before a binary operator of the shallowest parenthesis depth, and a string literal
that alone exceeds the limit is cut into adjacent literals.  Tokens are never changed, so the compiler sees the same program.
usage: reflow_cpp.py [--limit 160] [--check] FILE...   (--check: only report the lines over the limit, exit 1 if any)"""
import sys

TAB = 4


def width(s):
    return sum(TAB if ch == "\t" else 1 for ch in s)


def scan(code):
    """yield (index, char, paren depth, brace depth, in_literal) for every character; depth is the value BEFORE an opener / AFTER a closer"""
    out = []
    par = brc = 0
    i, n = 0, len(code)
    lit = None
    while i < n:
        ch = code[i]
        if lit:
            out.append((i, ch, par, brc, True))
            if ch == "\\":
                i += 1
                if i < n: out.append((i, code[i], par, brc, True))
            elif ch == lit:
                lit = None
            i += 1
            continue
        if ch in "\"'":
            # a digit separator (1'000) is not a literal
            if ch == "'" and i > 0 and code[i - 1].isalnum() and i + 1 < n and code[i + 1].isalnum() and not (i + 2 < n and code[i + 2] == "'"):
                out.append((i, ch, par, brc, False)); i += 1; continue
            lit = ch
            out.append((i, ch, par, brc, True)); i += 1; continue
        if ch in "([": out.append((i, ch, par, brc, False)); par += 1
        elif ch in ")]": par -= 1; out.append((i, ch, par, brc, False))
        elif ch == "{": out.append((i, ch, par, brc, False)); brc += 1
        elif ch == "}": brc -= 1; out.append((i, ch, par, brc, False))
        else: out.append((i, ch, par, brc, False))
        i += 1
    return out


def split_comment(line):
    sc = scan(line)
    for k, (i, ch, par, brc, lit) in enumerate(sc):
        if not lit and ch == "/" and i + 1 < len(line) and line[i + 1] == "/":
            return line[:i].rstrip(), line[i:]
    return line, ""


def wrap_comment(indent, comment, limit):
    words = comment[2:].split()
    lines, cur = [], indent + "//"
    for w in words:
        if width(cur + " " + w) > limit and cur.strip() != "//":
            lines.append(cur); cur = indent + "//"
        cur += " " + w
    lines.append(cur)
    return lines


def statements(body):
    """split `a; b; if(x) { c; } d;` at the semicolons / closing braces of depth 0"""
    sc = scan(body)
    parts, start = [], 0
    for k, (i, ch, par, brc, lit) in enumerate(sc):
        if lit or par != 0: continue
        if ch == ";" and brc == 0:
            parts.append(body[start:i + 1].strip()); start = i + 1
        elif ch == "}" and brc == 0:
            # a block ends here unless something that belongs to it follows (else / while of do / `;` of a lambda or struct / `)` / `,`)
            rest = body[i + 1:].lstrip()
            if rest.startswith(("else", ";", ")", ",", ".", "while")): continue
            parts.append(body[start:i + 1].strip()); start = i + 1
    tail = body[start:].strip()
    if tail: parts.append(tail)
    return [p for p in parts if p]


def find_block(code):
    """the first `{ ... }` on this line that holds statements: (open index, close index)"""
    sc = scan(code)
    stack = []
    for (i, ch, par, brc, lit) in sc:
        if lit: continue
        if ch == "{": stack.append(i)
        elif ch == "}" and stack:
            o = stack.pop()
            if not stack and ";" in code[o:i]:
                return o, i
    return None


def break_expression(indent, code, limit):
    """one statement, too long: break at the shallowest commas / binary operators, filling lines greedily"""
    sc = scan(code)
    cands = []     # (depth, position AFTER which the line may end)
    n = len(code)
    for (i, ch, par, brc, lit) in sc:
        if lit: continue
        depth = par + brc
        if ch == "," and i + 1 < n and code[i + 1] == " ": cands.append((depth, i + 1))
        elif ch in "+?" and 0 < i < n - 1 and code[i - 1] != ch and code[i + 1] != ch and code[i + 1] != "=" and code[i - 1] not in "(eE,=":
            cands.append((depth, i))            # break BEFORE the operator
        elif ch in "&|" and i + 1 < n and code[i + 1] == ch: cands.append((depth, i))
        elif ch == ":" and 0 < i < n - 1 and code[i - 1] == " " and code[i + 1] == " ": cands.append((depth, i))
        elif ch == ")" and i + 2 < n and code[i + 1] == " " and depth == 0 and code.lstrip().startswith(("if(", "for(", "while(", "else if(")):
            cands.append((-1, i + 1))          # `if(...) statement`: the statement goes to its own line first
    if not cands: return None
    for level in sorted(set(d for d, _ in cands)):
        pts = sorted(p for d, p in cands if d <= level)
        lines, start, cont = [], 0, indent
        ok = True
        while start < n:
            room = limit - width(cont)
            if width(code[start:]) <= room:
                lines.append(cont + code[start:].strip()); break
            best = None
            for p in pts:
                if p <= start: continue
                if width(code[start:p]) <= room: best = p
                else: break
            if best is None: ok = False; break
            lines.append(cont + code[start:best].strip()); start = best
            cont = indent + "\t"
        if ok: return lines
    return None


def cut_literal(indent, code, limit):
    sc = scan(code)
    room = limit - width(indent) - 2
    last_space = None
    col = 0
    for (i, ch, par, brc, lit) in sc:
        col += TAB if ch == "\t" else 1
        if col > room: break
        if lit and ch == " " and code[i - 1] != "\\": last_space = i
    if last_space is None: return None
    quote_open = code.rfind('"', 0, last_space)
    if quote_open < 0: return None
    return [indent + code[:last_space + 1] + '"', indent + "\t" + '"' + code[last_space + 1:]]


def reflow(line, limit, depth=0):
    if width(line) <= limit or depth > 12: return [line]
    stripped = line.lstrip("\t")
    indent = line[:len(line) - len(stripped)]
    if stripped.startswith("#") or line.rstrip().endswith("\\"): return [line]
    if stripped.startswith("//"): return wrap_comment(indent, stripped, limit)
    code, comment = split_comment(stripped)
    out = []
    if comment and code:
        out += wrap_comment(indent, comment, limit) if width(indent + comment) > limit else [indent + comment]
        return out + reflow(indent + code, limit, depth + 1)
    blk = find_block(code)
    if blk:
        o, cl = blk
        head, body, tail = code[:o + 1].rstrip(), code[o + 1:cl], code[cl:]
        res = reflow(indent + head, limit, depth + 1)
        for st in statements(body):
            res += reflow(indent + "\t" + st, limit, depth + 1)
        res += reflow(indent + tail.strip(), limit, depth + 1)
        return res
    sts = statements(code)
    if len(sts) > 1:
        res = []
        for st in sts: res += reflow(indent + st, limit, depth + 1)
        return res
    br = break_expression(indent, code, limit)
    if br:
        res = [br[0]]
        for l in br[1:]: res += reflow(l, limit, depth + 1) if width(l) > limit else [l]
        return res
    cut = cut_literal(indent, code, limit)
    if cut:
        return [cut[0]] + reflow(cut[1], limit, depth + 1)
    return [line]


def main():
    args = sys.argv[1:]
    limit, check = 160, False
    if "--limit" in args:
        k = args.index("--limit"); limit = int(args[k + 1]); del args[k:k + 2]
    if "--check" in args:
        check = True; args.remove("--check")
    bad = 0
    for path in args:
        src = open(path).read().split("\n")
        if check:
            for n, l in enumerate(src, 1):
                if len(l) > limit: print("%s:%d: %d characters" % (path, n, len(l))); bad += 1
            continue
        out = []
        for l in src: out += reflow(l, limit)
        if out != src: open(path, "w").write("\n".join(out))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
