#!/usr/bin/env python3
"""Generate cudaimageprocessing_amd/csrc/median5_network.h: the min/max networks of the 5x5
median kernel (median5_kernel in filters.hip).

The kernel sorts every 5-pixel column once (sort5), merges adjacent sorted columns pairwise
(merge55: two sorted 5-lists -> a sorted 10-list), and shares those between neighbouring outputs.
The median of a 5x5 window = the 13th smallest of 25 = rank 12 (0-based) of the union of two
merged column pairs A, B (10 values each) and one more sorted column C (5 values).  Since at least
7 and at most 12 values of A u B lie below it, only ranks 7..12 of A u B matter (ranks7to12),
and the median is the 6th smallest of those 6 values and the 5 of C (select6of11).

Networks: sort5 is the 9-exchange optimum; the merges are Batcher odd-even merges of sorted
lists, pruned by a backward liveness pass to the outputs that are used (half exchanges keep only
the min or the max).  Every network is verified with the 0-1 principle over all monotone inputs
(each sorted input list has len+1 binary patterns) before the header is written.
"""
import itertools
import os
import sys

SORT5 = [(0, 1), (3, 4), (2, 4), (2, 3), (1, 4), (0, 3), (0, 2), (1, 3), (1, 2)]


def oddeven_merge(a, b, ces):
    """Merge the sorted wire lists a and b; appends exchanges, returns the wires in sorted order."""
    if not a or not b:
        return a + b
    if len(a) == 1 and len(b) == 1:
        ces.append((a[0], b[0]))
        return [a[0], b[0]]
    even = oddeven_merge(a[0::2], b[0::2], ces)
    odd = oddeven_merge(a[1::2], b[1::2], ces)
    out = [even[0]]
    for i, o in enumerate(odd):
        if i + 1 < len(even):
            ces.append((o, even[i + 1]))
            out += [o, even[i + 1]]
        else:
            out.append(o)
    if len(even) > len(odd) + 1:
        out += even[len(odd) + 1:]
    return out


def liveness(ces, outs):
    live, kept = set(outs), []
    for a, b in reversed(ces):
        need_min, need_max = a in live, b in live
        if need_min or need_max:
            kept.append((a, b, need_min, need_max))
            live.add(a)
            live.add(b)
    kept.reverse()
    return kept


def monotone_patterns(groups):
    """Bit-parallel 0-1 inputs: wire masks over every combination of sorted binary groups."""
    pats = list(itertools.product(*[range(s + 1) for s in groups]))
    wires = [0] * sum(groups)
    ones = []
    for pi, ks in enumerate(pats):
        base = 0
        for gi, s in enumerate(groups):
            for r in range(s):
                if r >= s - ks[gi]:
                    wires[base + r] |= 1 << pi
            base += s
        ones.append(sum(ks))
    return wires, ones


def verify(groups, ops, where):
    """ops: [(a, b, need_min, need_max)]; where: {rank: wire}.  True if wire holds that rank."""
    w, ones = monotone_patterns(groups)
    n = sum(groups)
    for a, b, need_min, need_max in ops:
        lo, hi = w[a] & w[b], w[a] | w[b]
        if need_min:
            w[a] = lo
        if need_max:
            w[b] = hi
    for rank, wire in where.items():
        expect = 0
        for pi, c in enumerate(ones):
            if c >= n - rank:
                expect |= 1 << pi
        if w[wire] != expect:
            return False
    return True


def emit_ops(ops, indent="    "):
    lines = []
    for a, b, need_min, need_max in ops:
        if need_min and need_max:
            lines.append(f"{indent}{{ const T lo = mn(w{a}, w{b}); w{b} = mx(w{a}, w{b}); w{a} = lo; }}")
        elif need_min:
            lines.append(f"{indent}w{a} = mn(w{a}, w{b});")
        else:
            lines.append(f"{indent}w{b} = mx(w{a}, w{b});")
    return lines


def main():
    out = []
    stats = {}

    # ---- sort5: all 2^5 binary inputs
    full5 = [(a, b, True, True) for a, b in SORT5]
    assert verify([1] * 5, full5, {k: k for k in range(5)})
    stats["sort5"] = (len(SORT5), 2 * len(SORT5))

    # ---- merge55
    ces = []
    order55 = oddeven_merge(list(range(5)), list(range(5, 10)), ces)
    m55 = [(a, b, True, True) for a, b in ces]
    assert verify([5, 5], m55, {k: order55[k] for k in range(10)})
    stats["merge55"] = (len(ces), 2 * len(ces))

    # ---- ranks 7..12 of two sorted 10-lists
    ces = []
    order = oddeven_merge(list(range(10)), list(range(10, 20)), ces)
    part = liveness(ces, [order[k] for k in range(7, 13)])
    assert verify([10, 10], part, {k: order[k] for k in range(7, 13)})
    stats["ranks7to12"] = (len(part), sum(int(m) + int(x) for _, _, m, x in part))

    # ---- 6th smallest of a sorted 6-list s and a sorted 5-list c:
    #      min over i+j=6 of max(s_i, c_j) (1-based, s_0 = c_0 = -inf)
    def sel(s, c):
        terms = [max(s[i - 1], c[6 - i - 1]) for i in range(1, 6)] + [s[5]]
        return min(terms)
    for ks in itertools.product(range(7), range(6)):
        s = [0] * (6 - ks[0]) + [1] * ks[0]
        c = [0] * (5 - ks[1]) + [1] * ks[1]
        assert sel(s, c) == sorted(s + c)[5]
    stats["select6of11"] = (0, 10)

    # ---- whole construction on the monotone inputs of (A10, B10, C5): rank 12 of 25
    w, ones = monotone_patterns([10, 10, 5])
    for a, b, need_min, need_max in part:
        lo, hi = w[a] & w[b], w[a] | w[b]
        if need_min:
            w[a] = lo
        if need_max:
            w[b] = hi
    s = [w[order[k]] for k in range(7, 13)]
    c = w[20:25]
    terms = [s[i - 1] | c[6 - i - 1] for i in range(1, 6)] + [s[5]]
    med = terms[0]
    for t in terms[1:]:
        med &= t
    expect = 0
    for pi, cnt in enumerate(ones):
        if cnt >= 25 - 12:
            expect |= 1 << pi
    assert med == expect, "median construction failed the 0-1 check"

    hdr = [
        "// median5_network.h -- GENERATED by tools/gen_median5_network.py; do not edit.",
        "// Min/max networks of the 5x5 median kernel; every one verified by the generator with the",
        "// 0-1 principle over all monotone inputs.  T is any type, mn/mx its min and max.",
    ]
    for k, (c, o) in stats.items():
        hdr.append(f"//   {k}: {c} exchanges, {o} min/max operations" if c else f"//   {k}: {o} min/max operations")
    hdr += ["#pragma once", ""]
    out += hdr

    out += ["// v[0..4] -> ascending", "template <class T, class MN, class MX>",
            "MI_DEV void sort5(T (&v)[5], MN mn, MX mx)", "{",
            "    T w0 = v[0], w1 = v[1], w2 = v[2], w3 = v[3], w4 = v[4];"]
    out += emit_ops(full5)
    out += ["    v[0] = w0; v[1] = w1; v[2] = w2; v[3] = w3; v[4] = w4;", "}", ""]

    out += ["// a, b ascending -> m = their union, ascending", "template <class T, class MN, class MX>",
            "MI_DEV void merge55(const T (&a)[5], const T (&b)[5], T (&m)[10], MN mn, MX mx)", "{",
            "    T " + ", ".join(f"w{i} = a[{i}]" for i in range(5)) + ";",
            "    T " + ", ".join(f"w{i + 5} = b[{i}]" for i in range(5)) + ";"]
    out += emit_ops(m55)
    out += ["    " + " ".join(f"m[{k}] = w{order55[k]};" for k in range(10)), "}", ""]

    out += ["// a, b ascending 10-lists -> s = ranks 7..12 (0-based) of their union, ascending",
            "template <class T, class MN, class MX>",
            "MI_DEV void ranks7to12(const T (&a)[10], const T (&b)[10], T (&s)[6], MN mn, MX mx)", "{"]
    used = set()
    for a, b, _, _ in part:
        used.add(a)
        used.add(b)
    used |= {order[k] for k in range(7, 13)}
    for i in sorted(used):
        src = f"a[{i}]" if i < 10 else f"b[{i - 10}]"
        out.append(f"    T w{i} = {src};")
    out += emit_ops(part)
    out += ["    " + " ".join(f"s[{k - 7}] = w{order[k]};" for k in range(7, 13)), "}", ""]

    out += ["// s ascending 6-list, c ascending 5-list -> the 6th smallest of their union",
            "template <class T, class MN, class MX>",
            "MI_DEV T select6of11(const T (&s)[6], const T (&c)[5], MN mn, MX mx)", "{",
            "    T r = s[5];"]
    for i in range(1, 6):
        out.append(f"    r = mn(r, mx(s[{i - 1}], c[{6 - i - 1}]));")
    out += ["    return r;", "}", ""]

    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.normpath(os.path.join(here, "..", "cudaimageprocessing_amd", "csrc", "median5_network.h"))
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)
    for k, v in stats.items():
        print(f"  {k}: {v[0]} exchanges, {v[1]} min/max ops")


if __name__ == "__main__":
    sys.exit(main())
