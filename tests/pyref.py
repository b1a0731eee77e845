"""Independent pure-Python statement of the scan's CLOSED FORM, used to cross-check the
C oracle on random strings.  Small inputs only.  Test infrastructure."""

FWD = {c: v for c, v in zip("ACGTUacgtu", [0, 1, 2, 3, 3] * 2)}


def code(ch, comp=False):
    v = FWD.get(ch)
    if v is None:
        return 0            # non-ACGTU packs as 0 in either orientation (4 & 3)
    return 3 - v if comp else v


def pack(codes):
    x = 0
    for i, c in enumerate(codes):
        x |= (c & 3) << (2 * i)
    return x


def scan_closed_form(seq, k, m):
    """-> list of (base, length, read_offset, flag, ctx0, ctx1); seq is a str of single-byte chars.
    A run [s,e] of identical bytes is recorded iff n >= max(m,2), s >= k, e + k <= L-1.  A qualifying run of a
    non-ACGTU byte re-uses (ctx, base, flag) of the previous recorded ACGTU tract in the read, or is dropped when
    there is none (undefined behaviour in the reference)."""
    L = len(seq)
    mm = max(m, 2)
    out = []
    last = None
    s = 0
    while s < L:
        e = s
        while e + 1 < L and seq[e + 1] == seq[s]:
            e += 1
        n = e - s + 1
        if n >= mm and s >= k and e + k <= L - 1:
            ch = seq[s]
            if ch in FWD:
                left = seq[s - k:s]
                right = seq[e + 1:e + 1 + k]
                f = FWD[ch]
                if f < 2:
                    last = (f, 1, pack([code(c) for c in left]), pack([code(c) for c in right]))
                else:
                    last = (3 - f, 2, pack([code(c, True) for c in reversed(right)]),
                            pack([code(c, True) for c in reversed(left)]))
            if last is not None:
                out.append((last[0], n, s - k, last[1], last[2], last[3]))
        s = e + 1
    return out


def scan_all_monomers(seq, k):
    """reference: update_hopo_counter_from_seq_all_monomers (src/hopo_counter.c:260-283): every position k <= i < L-k
    whose byte differs from both neighbours is recorded with length 1; a non-ACGTU byte re-uses the previous record's
    context (or is dropped when there is none)."""
    L = len(seq)
    out = []
    last = None
    for i in range(k, L - k):
        if seq[i] == seq[i - 1] or seq[i] == seq[i + 1]:
            continue
        ch = seq[i]
        if ch in FWD:
            left = seq[i - k:i]
            right = seq[i + 1:i + 1 + k]
            f = FWD[ch]
            if f < 2:
                last = (f, 1, pack([code(c) for c in left]), pack([code(c) for c in right]))
            else:
                last = (3 - f, 2, pack([code(c, True) for c in reversed(right)]),
                        pack([code(c, True) for c in reversed(left)]))
        if last is not None:
            out.append((last[0], 1, i - k, last[1], last[2], last[3]))
    return out


def figure2_reads(fig2, both_strands=True):
    """The reads behind the reference's second figure (recipe/200322_002.png, tests/golden/known_answers.json "figure2"):
    for every drawn tract length n, `count` reads left + base^n + right -- half of them as the reverse complement when
    `both_strands` (the figure does not say which strand a read came from; both store the same context)."""
    comp = str.maketrans("ACGT", "TGCA")
    reads = []
    for n, cnt in sorted(fig2["reads_per_length"].items(), key=lambda t: int(t[0])):
        fwd = fig2["left"] + fig2["base"] * int(n) + fig2["right"]
        rev = fwd.translate(comp)[::-1]
        for j in range(cnt):
            reads.append(rev if (both_strands and j % 2) else fwd)
    return reads


def figure2_expected(fig2):
    """(length, count) of the finalised elements in the reference's order (length descending inside the one context,
    src/hopo_counter.c:28-38) and of the context's length histogram (highest count first, src/context_histogram.c:278-286)"""
    bars = [(int(n), c) for n, c in fig2["reads_per_length"].items()]
    return sorted(bars, key=lambda t: -t[0]), sorted(bars, key=lambda t: (-t[1], -t[0]))
