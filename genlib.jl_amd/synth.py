"""Deterministic synthetic pedigrees of the shapes BASELINE.json names.

Integer-only and counter-based (SplitMix64 of seed + k*gamma), so the same pedigree can be
regenerated bit-for-bit in any language (the Julia baseline reads the TSV written by
`write_tsv`, the reference's own on-disk format: header `ind father mother sex`, tabs,
0 = unknown parent; src/create.jl:161-189).

Draw k (0-based) of the stream is splitmix64_mix(seed + (k + 1) * GAMMA).  Individual with
1-based id i owns draws 8*(i-1) .. 8*(i-1)+7:
    +0 founder?        (u % 1000 < 20  -> no parents)            [random-mating only]
    +1 father from g-2 (u % 1000 < skip_permille, g >= 2)        [random-mating only]
    +2 father pick     (u % number_of_candidates)
    +3 mother from g-2 (u % 1000 < skip_permille, g >= 2)        [random-mating only]
    +4 mother pick
Generation-level draws of the deep consanguineous shape use 8*n_ind + 8*g + s.
"""
import numpy as np

SEED = 20241016
_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, k):
    """k-th output (vectorised over k, 0-based) of SplitMix64 seeded with `seed`."""
    k = np.asarray(k, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (k + np.uint64(1)) * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        return z ^ (z >> np.uint64(31))


def random_mating(n_ind, n_pro, n_gen, seed=SEED, skip_permille=0):
    """cfg3 / cfg4 shape: G discrete generations, last one = the n_pro probands.

    skip_permille > 0 draws a parent from generation g-2 instead of g-1 with that
    probability (overlapping generations -> "dragged" individuals in the cuts).  SURVEY.md
    8(d) proposed 50 for both synthetic configurations.  At 1e6 individuals (cfg4) that makes
    every deep ancestor reachable at many parent-step distances and the cuts grow to 2e5 members
    (3.7 TB of algorithmic traffic, two level matrices > 288 GB) instead of the ~3e4 / 260 GB
    the same section and BASELINE.md size the configuration at: the headline workload `cfg4`
    uses 0, `cfg4o` 5, and the literal one is the column-panel case budgeted in DESIGN.md 6.2.
    At 1e5 individuals (cfg3) the literal 50 fits one GPU and IS a bench workload and a
    full-size oracle test (`cfg3s`: cuts to 20,540 members, most of them dragged along;
    DESIGN.md 4.3c); `cfg3` is the same shape with 0.  Small skip > 0 pedigrees, genea140 and
    geneaJi exercise the dragged path in the parity tests.

    Returns (ind, father, mother, sex, proband_ids); ids are 1..n_ind in generation order
    (parents always have smaller ids).  Sex alternates 1, 2 inside a generation.
    """
    G = int(n_gen)
    per = (n_ind - n_pro) // (G - 1)
    sizes = [per] * (G - 1) + [n_pro]
    sizes[0] += (n_ind - n_pro) - per * (G - 1)
    starts = np.concatenate([[0], np.cumsum(sizes)])          # 0-based offset of each generation
    ind = np.arange(1, n_ind + 1, dtype=np.int64)
    gen = np.repeat(np.arange(G), sizes)
    local = np.arange(n_ind) - starts[gen]
    sex = (local % 2 + 1).astype(np.int64)
    base = (np.arange(n_ind, dtype=np.uint64) * np.uint64(8))
    u = [splitmix64(seed, base + np.uint64(d)) for d in range(5)]
    is_founder = (gen == 0) | ((u[0] % np.uint64(1000)) < np.uint64(20))

    def pick(sexcode, from2, upick):
        # candidates of sex `sexcode` in generation g-1 (or g-2): local index 2*t + (sexcode-1)
        src = np.where((from2 % np.uint64(1000) < np.uint64(skip_permille)) & (gen >= 2), gen - 2, np.maximum(gen - 1, 0))
        cnt = (np.asarray(sizes)[src] + (2 - sexcode)) // 2       # males: ceil(n/2), females: floor(n/2)
        cnt = np.maximum(cnt, 1)
        t = (upick % cnt.astype(np.uint64)).astype(np.int64)
        return starts[src] + 2 * t + (sexcode - 1) + 1             # 1-based id

    father = np.where(is_founder, 0, pick(1, u[1], u[2])).astype(np.int64)
    mother = np.where(is_founder, 0, pick(2, u[3], u[4])).astype(np.int64)
    pro = ind[starts[G - 1]:].copy()
    return ind, father, mother, sex, pro


def deep_inbred(n_gen=200, per_gen=50, n_sires=3, seed=SEED):
    """cfg5 shape: n_gen generations of per_gen (half male / half female); each generation
    uses n_sires sires drawn from the previous generation's males, mothers uniform among its
    females.  Probands = last generation."""
    n_ind = n_gen * per_gen
    ind = np.arange(1, n_ind + 1, dtype=np.int64)
    gen = np.arange(n_ind) // per_gen
    local = np.arange(n_ind) % per_gen
    sex = (local % 2 + 1).astype(np.int64)
    nm, nf = (per_gen + 1) // 2, per_gen // 2
    base = np.arange(n_ind, dtype=np.uint64) * np.uint64(8)
    u_f = splitmix64(seed, base + np.uint64(2))
    u_m = splitmix64(seed, base + np.uint64(4))
    gbase = np.uint64(8 * n_ind) + np.arange(n_gen, dtype=np.uint64) * np.uint64(8)
    sires = np.stack([(splitmix64(seed, gbase + np.uint64(s)) % np.uint64(nm)).astype(np.int64)
                      for s in range(n_sires)], axis=1)          # [g, s] local male index in g-1
    which = (u_f % np.uint64(n_sires)).astype(np.int64)
    prev0 = (np.maximum(gen - 1, 0)) * per_gen
    father = prev0 + 2 * sires[gen, which] + 1
    mother = prev0 + 2 * (u_m % np.uint64(nf)).astype(np.int64) + 1 + 1
    father = np.where(gen == 0, 0, father).astype(np.int64)
    mother = np.where(gen == 0, 0, mother).astype(np.int64)
    pro = ind[(n_gen - 1) * per_gen:].copy()
    return ind, father, mother, sex, pro


def chain_two_lines(depth):
    """Two probands descending from one founder couple through `depth` single-parent-known
    generations each: kinship 2^-(2*depth+1)-ish, reaching Float32 subnormals for depth ~ 70
    (the rare-branch test: subnormal stores must round exactly like the reference)."""
    ind, father, mother, sex = [1, 2], [0, 0], [0, 0], [1, 2]
    nxt = 3
    tips = []
    for _ in range(2):
        parent_f, parent_m = 1, 2
        for d in range(depth):
            ind.append(nxt); father.append(parent_f); mother.append(parent_m); sex.append(1)
            parent_f, parent_m = nxt, 0
            nxt += 1
        tips.append(nxt - 1)
    a = lambda x: np.asarray(x, dtype=np.int64)
    return a(ind), a(father), a(mother), a(sex), a(tips)


def parents_first_shuffle(ind, father, mother, sex, seed=1):
    """A random file order in which every parent still precedes its children but the depths are
    interleaved (what a hand-maintained pedigree file looks like): the input of
    gen.genealogy(...; sort=false), where the rank is the file position (src/create.jl:131,161,
    :234-254) and need not follow the ancestral depth.  Visits the individuals in a random order and
    places each one right after its not-yet-placed ancestors."""
    ind = np.asarray(ind, dtype=np.int64)
    n = len(ind)
    pos = {int(i): k for k, i in enumerate(ind)}
    fa = [pos[int(x)] if x else -1 for x in father]
    mo = [pos[int(x)] if x else -1 for x in mother]
    placed = np.zeros(n, dtype=bool)
    out = []
    for start in np.random.default_rng(seed).permutation(n):
        stack = [int(start)]
        while stack:
            x = stack[-1]
            if placed[x]:
                stack.pop()
                continue
            todo = [q for q in (fa[x], mo[x]) if q >= 0 and not placed[q]]
            if todo:
                stack.extend(todo)
            else:
                placed[x] = True
                out.append(x)
                stack.pop()
    o = np.asarray(out, dtype=np.int64)
    return ind[o], np.asarray(father, dtype=np.int64)[o], np.asarray(mother, dtype=np.int64)[o], np.asarray(sex, dtype=np.int64)[o]


def write_tsv(path, ind, father, mother, sex):
    """Reference on-disk format (data/geneaJi.csv): tab-separated, header row."""
    with open(path, "w") as fh:
        fh.write("ind\tfather\tmother\tsex\n")
        np.savetxt(fh, np.stack([ind, father, mother, sex], axis=1), fmt="%d", delimiter="\t")
