"""gfa_canon -- canonical form of a gbuilder GFA (the parity contract of SURVEY.md section 8a).

The reference's GFA text is not byte-reproducible across thread counts (segment
ids follow the XXH3 bucket order of 10*T buckets, link order follows BooPHF
indices), so two GFAs are compared through this canonical form:

 1. a circular segment (first k bases == last k bases) is replaced by the
    lexicographically largest rotation of the circle or of its reverse
    complement, re-linearised with k bases of overlap;
 2. segments are sorted and renamed to their rank;
 3. every link (a, oa, b, ob) is rewritten over ranks and replaced by
    min((a,oa,b,ob), (b,!ob,a,!oa)); links are sorted and de-duplicated;
 4. KC tags are kept per segment (optional).

Usage: python -m spades_for_blackbird_amd.tools.gfa_canon graph.gfa [-k K] [--kc]
"""
import hashlib
import sys

_COMP = str.maketrans("ACGT", "TGCA")


def rc(s):
    return s[::-1].translate(_COMP)


def _canon_circle(s, k):
    circ = s[: len(s) - k]
    best = None
    for c in (circ, rc(circ)):
        dbl = c + c
        n = len(c)
        for i in range(n):
            cand = dbl[i:i + n]
            if best is None or cand > best:
                best = cand
    # re-linearise: append the first k bases of the rotated circle
    ext = best
    while len(ext) < len(best) + k:
        ext += best
    return ext[: len(best) + k]


def canon(text, k=None, with_kc=False):
    """Returns (segments, links) : segments = sorted list of (seq[, kc]); links = sorted list of tuples."""
    segs = {}
    links = []
    for line in text.splitlines():
        if not line:
            continue
        f = line.split("\t")
        if f[0] == "S":
            kc = None
            for tag in f[3:]:
                if tag.startswith("KC:i:"):
                    kc = int(tag[5:])
            segs[f[1]] = (f[2], kc)
        elif f[0] == "L":
            if k is None:
                k = int(f[5][:-1])
            links.append((f[1], f[2], f[3], f[4]))
    if k is None:
        raise ValueError("k unknown: no L lines and no -k given")
    new_seq = {}
    flipped = {}
    # a perfect loop touches nothing but itself, in one orientation (a junction-free cycle);
    # a unitig that merely starts and ends at the same junction k-mer is not rotated
    not_loop = set()
    for a, oa, b, ob in links:
        if a != b or oa != ob:
            not_loop.add(a)
            not_loop.add(b)
    for name, (s, kc) in segs.items():
        flipped[name] = False
        if len(s) > k and s[:k] == s[len(s) - k:] and name not in not_loop:
            c = _canon_circle(s, k)
            new_seq[name] = c
        else:
            r = rc(s)
            if s < r:  # reference keeps s >= rc(s); tolerate either
                new_seq[name] = r
                flipped[name] = True
            else:
                new_seq[name] = s
    order = sorted(segs, key=lambda n: new_seq[n])
    rank = {n: i for i, n in enumerate(order)}
    out_segs = [(new_seq[n], segs[n][1]) if with_kc else (new_seq[n],) for n in order]
    out_links = set()

    def flip(o):
        return "-" if o == "+" else "+"

    for a, oa, b, ob in links:
        if flipped[a]:
            oa = flip(oa)
        if flipped[b]:
            ob = flip(ob)
        x = (rank[a], oa, rank[b], ob)
        y = (rank[b], flip(ob), rank[a], flip(oa))
        out_links.add(min(x, y))
    return out_segs, sorted(out_links)


def canon_text(text, k=None, with_kc=False):
    segs, links = canon(text, k, with_kc)
    lines = []
    for i, s in enumerate(segs):
        lines.append("S\t%d\t%s" % (i, s[0]) + ("\tKC:i:%d" % s[1] if with_kc and s[1] is not None else ""))
    for a, oa, b, ob in links:
        lines.append("L\t%d\t%s\t%d\t%s" % (a, oa, b, ob))
    return "\n".join(lines) + "\n"


def canon_md5(text, k=None, with_kc=False):
    return hashlib.md5(canon_text(text, k, with_kc).encode()).hexdigest()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    k = None
    with_kc = False
    paths = []
    i = 0
    while i < len(argv):
        if argv[i] == "-k":
            k = int(argv[i + 1])
            i += 2
        elif argv[i] == "--kc":
            with_kc = True
            i += 1
        else:
            paths.append(argv[i])
            i += 1
    for p in paths:
        with open(p) as f:
            print(canon_md5(f.read(), k, with_kc), p)
    return 0


if __name__ == "__main__":
    sys.exit(main())
