"""Synthetic inputs of the bench workload and of the full-size parity tests (C2 assembly, C3 coverage), built on the
device with torch so that 3 Gbp take seconds.  Seeded: the same bytes in every process (bench.py ranks, tests, tools).
Not part of the product path: nothing under csrc/ or cli/ uses it."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def contig_lengths(total_target):
    """hifiasm-like contig lengths: the reference's own HG002 assembly BED fixture (100 contigs, 3.16 Gb,
    largest 242 Mb), data file of test/bigenough/hg002-cornetto-E_3 kept under tests/golden/."""
    path = os.path.join(ROOT, "tests", "golden", "bigenough", "chroms.bed")
    lens = [int(l.split()[2]) for l in open(path) if l.strip()]
    if total_target and total_target < sum(lens):
        scale = total_target / float(sum(lens))
        lens = [max(1000, int(x * scale)) for x in lens]
    return lens


def _plant(torch, dev, bases, starts, lengths, units, unit_ids):
    """write tandem repeats: feature j = units[unit_ids[j]] repeated over bases[starts[j] : starts[j] + lengths[j]]
    (all features at once on the device; lengths <= 512)"""
    if len(starts) == 0:
        return
    maxlen = int(max(lengths))
    umax = max(len(u) for u in units)
    utab = torch.zeros((len(units), umax), dtype=torch.uint8)
    ulen = torch.zeros(len(units), dtype=torch.int64)
    for i, u in enumerate(units):
        utab[i, :len(u)] = torch.frombuffer(bytearray(u), dtype=torch.uint8)
        ulen[i] = len(u)
    utab, ulen = utab.to(dev), ulen.to(dev)
    st = torch.from_numpy(np.asarray(starts, dtype=np.int64)).to(dev)
    ln = torch.from_numpy(np.asarray(lengths, dtype=np.int64)).to(dev)
    ui = torch.from_numpy(np.asarray(unit_ids, dtype=np.int64)).to(dev)
    step = 1 << 16
    ar = torch.arange(maxlen, device=dev)
    for s in range(0, len(starts), step):
        e = min(len(starts), s + step)
        idx = st[s:e, None] + ar[None, :]
        ok = ar[None, :] < ln[s:e, None]
        val = utab[ui[s:e, None], ar[None, :] % ulen[ui[s:e], None]]
        bases[idx[ok]] = val[ok]


def _revcomp_np(a):
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    return comp[a[::-1]]


def _humanlike_base(torch, dev, total, seed, g):
    """the background of the humanlike profile: isochores of 100 kb - 1 Mb (log-uniform) whose GC content is drawn from 35-55 %,
    bases independent inside an isochore, then CpG taken down to ~20 % of its expectation the way genomes lose it (80 % of the CpG
    dinucleotides deaminated: CG -> TG or, for the other strand, CG -> CA) — which also gives the TpG / CpA excess of a real genome"""
    rs = np.random.default_rng(seed ^ 0x150C40)
    n_iso = int(total / 2.0e5) + 16
    iso_len = (10 ** rs.uniform(5.0, 6.0, size=n_iso)).astype(np.int64)
    while int(iso_len.sum()) < total:
        iso_len = np.concatenate([iso_len, (10 ** rs.uniform(5.0, 6.0, size=n_iso)).astype(np.int64)])
    bounds = torch.from_numpy(np.cumsum(iso_len)).to(dev)
    gc = torch.from_numpy(rs.uniform(0.35, 0.55, size=len(iso_len)).astype(np.float32)).to(dev)
    bases = torch.empty(total, dtype=torch.uint8, device=dev)
    step = 1 << 27
    at = torch.tensor(list(b"AT"), dtype=torch.uint8, device=dev)
    cg = torch.tensor(list(b"CG"), dtype=torch.uint8, device=dev)
    for s0 in range(0, total, step):
        e0 = min(total, s0 + step)
        pos = torch.arange(s0, e0, device=dev)
        p_gc = gc[torch.bucketize(pos, bounds, right=True).clamp_(max=len(iso_len) - 1)]
        del pos
        is_gc = torch.rand(e0 - s0, device=dev, generator=g) < p_gc
        del p_gc
        bit = torch.randint(0, 2, (e0 - s0,), device=dev, generator=g)
        bases[s0:e0] = torch.where(is_gc, cg[bit], at[bit])
        del is_gc, bit
    for s0 in range(0, total - 1, step):                # CpG depletion (a chunk's last base pairs with the next chunk's first)
        e0 = min(total - 1, s0 + step)
        cpg = (bases[s0:e0] == 67) & (bases[s0 + 1:e0 + 1] == 71)
        r = torch.rand(e0 - s0, device=dev, generator=g)
        c2t = cpg & (r < 0.4)
        g2a = cpg & (r >= 0.4) & (r < 0.8)
        del cpg, r
        bases[s0:e0][c2t] = 84
        bases[s0 + 1:e0 + 1][g2a] = 65
        del c2t, g2a
    return bases


def _plant_copies(torch, dev, bases, starts, lens_, cons_off, cons, flip, div, g):
    """diverged copies of (a part of) a consensus: copy j = cons[cons_off[j] : cons_off[j] + lens_[j]] (its reverse complement when
    flip[j]), every base substituted with probability div[j], written at bases[starts[j] ...] (ragged, batched on the device)"""
    if len(starts) == 0:
        return
    # copies that would overlap an earlier one (by start) are dropped: a scatter with colliding destinations has no defined winner,
    # and the assembly must be the same bytes in every process
    starts, lens_, cons_off = np.asarray(starts, np.int64), np.asarray(lens_, np.int64), np.asarray(cons_off, np.int64)
    flip, div = np.asarray(flip, bool), np.asarray(div, np.float32)
    order = np.argsort(starts, kind="stable")
    starts, lens_, cons_off, flip, div = starts[order], lens_[order], cons_off[order], flip[order], div[order]
    keep = np.ones(len(starts), bool)
    end = -1
    for j in range(len(starts)):
        if starts[j] < end:
            keep[j] = False
        else:
            end = starts[j] + lens_[j]
    starts, lens_, cons_off, flip, div = starts[keep], lens_[keep], cons_off[keep], flip[keep], div[keep]
    cons_f = torch.from_numpy(np.ascontiguousarray(cons)).to(dev)
    cons_r = torch.from_numpy(np.ascontiguousarray(_revcomp_np(cons))).to(dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    n_cons = len(cons)
    csum = np.concatenate([[0], np.cumsum(lens_)])
    budget, i = 1 << 25, 0                               # elements per batch
    while i < len(starts):
        j = int(np.searchsorted(csum, csum[i] + budget, "right")) - 1
        j = max(j, i + 1)
        ln = torch.from_numpy(lens_[i:j]).to(dev)
        tot = int(csum[j] - csum[i])
        rep = torch.repeat_interleave(torch.arange(j - i, device=dev), ln, output_size=tot)
        within = torch.arange(tot, device=dev) - torch.from_numpy(csum[i:j] - csum[i]).to(dev)[rep]
        dst = torch.from_numpy(starts[i:j]).to(dev)[rep] + within
        co = torch.from_numpy(cons_off[i:j]).to(dev)[rep]
        fl = torch.from_numpy(flip[i:j]).to(dev)[rep]
        # a flipped copy reads the reverse complement of the same stretch of the consensus
        src_f = co + within
        src_r = (n_cons - co - ln[rep]) + within
        val = torch.where(fl, cons_r[src_r.clamp_(0, n_cons - 1)], cons_f[src_f.clamp_(0, n_cons - 1)])
        mut = torch.rand(tot, device=dev, generator=g) < torch.from_numpy(div[i:j]).to(dev)[rep]
        val = torch.where(mut, lut[torch.randint(0, 4, (tot,), device=dev, generator=g)], val)
        bases[dst] = val
        del rep, within, dst, co, fl, src_f, src_r, val, mut
        i = j


def make_assembly(torch, dev, lens, seed, profile="uniform"):
    """bases (uint8 ASCII, contigs at 64-byte aligned offsets) with planted features — SURVEY 8d, C2.
    profile "satellite" additionally plants what a real human assembly is full of: HSat2/3-like (CATTC)n / (GGAAT)n
    arrays of 0.1-5 Mb (half of them exact, half with 2 % substitutions) over >= 3 % of the bases, (AT)n / (AAAG)n
    microsatellites every ~20 kb and poly-A / poly-T runs every ~10 kb.
    profile "humanlike" has those on a background with the COMPOSITION of a human assembly instead of uniform bases (what the
    position-parallel sieve of sd_sift is sensitive to): isochores with 35-55 % GC, CpG at ~20 % of its expectation, ~10 % of the
    bases in 85-95 %-identity copies of a 300-bp Alu-like consensus with 10-40-base poly-A tails, ~15 % in 5'-truncated
    80-95 %-identity copies of an AT-rich 6 kb L1-like consensus."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    offs, pos = [], 0
    for n in lens:
        offs.append(pos)
        pos = (pos + n + 63) // 64 * 64
    total = pos + 256
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    if profile == "humanlike":
        bases = _humanlike_base(torch, dev, total, seed, g)
    else:
        codes = torch.randint(0, 4, (total,), dtype=torch.uint8, device=dev, generator=g)
        bases = lut[codes.long()] if total < (1 << 28) else None
        if bases is None:                                  # chunked lookup keeps the int64 index temporary small
            bases = torch.empty_like(codes)
            step = 1 << 28
            for s in range(0, total, step):
                bases[s:s + step] = lut[codes[s:s + step].long()]
        del codes
    rng = np.random.default_rng(seed)

    def put(p, b):
        bases[p:p + len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)

    if profile == "humanlike":
        rh = np.random.default_rng(seed ^ 0xA1B2C3)
        acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
        # L1-like: 6 kb, 58 % AT, ends in a poly-A tail; copies are 5'-truncated (they keep the 3' end), 80-95 % identity
        l1 = acgt[rh.choice(4, size=6000, p=[0.33, 0.21, 0.21, 0.25])].copy()
        l1[-30:] = ord("A")
        # Alu-like: two GC-rich arms around an A-rich linker, then the poly-A tail (drawn per copy: 10-40)
        alu = acgt[rh.choice(4, size=340, p=[0.21, 0.30, 0.31, 0.18])].copy()
        alu[120:135] = np.frombuffer(b"AAAAATACAAAAAAT"[:15], dtype=np.uint8)
        alu[300:] = ord("A")
        big = [(o, n) for o, n in zip(offs, lens) if n >= 20000]
        wts = np.array([n for _, n in big], dtype=np.float64)
        wts /= wts.sum()
        nb = float(sum(n for _, n in big))

        def scatter(count, max_len):
            ci = rh.choice(len(big), size=count, p=wts)
            o = np.array([big[i][0] for i in ci], dtype=np.int64)
            n = np.array([big[i][1] for i in ci], dtype=np.int64)
            return o + 2000 + (rh.random(count) * (n - max_len - 4000)).astype(np.int64)

        n_l1 = int(0.17 * nb / 1050.0)                     # (about a tenth of them overlap an earlier one and are dropped)                     # mean fragment ~1.05 kb (log-uniform 100 .. 6000)
        ln = np.minimum(6000, (10 ** rh.uniform(2.0, np.log10(6000.0), size=n_l1)).astype(np.int64))
        _plant_copies(torch, dev, bases, scatter(n_l1, 6000), ln, 6000 - ln, l1, rh.random(n_l1) < 0.5, rh.uniform(0.05, 0.20, size=n_l1), g)
        n_alu = int(0.11 * nb / 325.0)
        ln = 300 + rh.integers(10, 41, size=n_alu)
        _plant_copies(torch, dev, bases, scatter(n_alu, 340), ln, np.zeros(n_alu, np.int64), alu, rh.random(n_alu) < 0.5, rh.uniform(0.05, 0.15, size=n_alu), g)
    if profile in ("satellite", "humanlike"):
        rs = np.random.default_rng(seed ^ 0x5A7E111)
        # microsatellites and homopolymer runs, everywhere
        st, ln, ui = [], [], []
        units = [b"AT", b"AAAG", b"A", b"T", b"CA", b"TTTC"]
        for off, n in zip(offs, lens):
            if n < 50000:
                continue
            p = np.arange(7000, n - 2000, 20000) + rs.integers(0, 4000, size=len(np.arange(7000, n - 2000, 20000)))
            st += (off + p).tolist(); ln += rs.integers(20, 121, size=len(p)).tolist(); ui += rs.choice([0, 1, 4, 5], size=len(p)).tolist()
            p = np.arange(3000, n - 2000, 10000) + rs.integers(0, 2000, size=len(np.arange(3000, n - 2000, 10000)))
            st += (off + p).tolist(); ln += rs.integers(12, 41, size=len(p)).tolist(); ui += rs.choice([2, 3], size=len(p)).tolist()
        _plant(torch, dev, bases, st, ln, units, ui)
        # satellite arrays: log-uniform 0.1-5 Mb, in the larger contigs, until 3.2 % of the bases are covered
        want = int(0.032 * sum(lens))
        have, k = 0, 0
        big = [i for i in range(len(lens)) if lens[i] >= 12_000_000] or [int(np.argmax(lens))]
        slots = {}
        while have < want:
            ci = big[k % len(big)]
            L = int(min(10 ** rs.uniform(5.0, 6.7), lens[ci] // 8))
            nth = slots.get(ci, 0)
            slots[ci] = nth + 1
            p = int(lens[ci] * (0.15 + 0.1 * nth) % (lens[ci] - L - 100000)) + 50000
            unit = (b"CATTC", b"GGAAT")[k % 2]
            arr = torch.frombuffer(bytearray(unit), dtype=torch.uint8).to(dev).repeat(L // 5 + 1)[:L].clone()
            if k % 4 >= 2:                             # diverged copy: 2 % substitutions
                m = torch.rand(L, device=dev, generator=g) < 0.02
                arr[m] = lut[torch.randint(0, 4, (int(m.sum()),), device=dev, generator=g)]
            bases[offs[ci] + p: offs[ci] + p + L] = arr
            have += L
            k += 1
    for off, n in zip(offs, lens):
        if n < 50000:
            continue
        put(off, b"CCCTAA" * 2000)
        put(off + n - 9000, b"TTAGGG" * 1500)
        k = 0
        for p in range(500000, n - 20000, 500000):
            kind = k % 4
            k += 1
            if kind == 0:
                put(off + p, b"TTAGGG" * int(rng.integers(3, 81)))
            elif kind == 1:
                put(off + p, bytes([b"ACGT"[int(rng.integers(0, 4))]]) * int(rng.integers(10, 301)))
            elif kind == 2:
                u = bytes(b"ACGT"[int(x)] for x in rng.integers(0, 4, size=2))
                put(off + p, u * int(rng.integers(10, 201)))
            else:
                put(off + p, b"N" * int(rng.integers(1, 501)))
        lo = off + n // 2
        bases[lo:lo + 500] |= 0x20                     # one 500-bp lower-case stretch
    return bases, np.array(offs, dtype=np.int64)


def make_coverage(torch, dev, lens, offs, seed):
    """per-base depth / mq-depth (u16 stored as int16 bit patterns) — SURVEY 8d, C3"""
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 1)
    total = int(offs[-1] + (lens[-1] + 63) // 64 * 64 + 256)
    nk = (total + 999) // 1000
    base = torch.poisson(torch.full((nk,), 30.0, device=dev), generator=g).to(torch.int16)
    depth = base.repeat_interleave(1000)[:total].contiguous()
    del base
    step = 1 << 28
    for s in range(0, total, step):
        e = min(total, s + step)
        depth[s:e] += torch.randint(-2, 3, (e - s,), dtype=torch.int16, device=dev, generator=g)
    depth.clamp_(min=0)
    mq = depth.clone()
    rng = np.random.default_rng(seed + 1)
    for off, n in zip(offs, lens):
        k = 0
        for p in range(400000, n - 70000, 400000):
            L = int(rng.integers(2000, 60001))
            s = int(off) + p
            if k % 2 == 0:
                depth[s:s + L] //= 5
            else:
                depth[s:s + L] *= 3
            mq[s:s + L] = depth[s:s + L]
            k += 1
        for p in range(500000, n - 70000, 500000):
            L = int(rng.integers(2000, 60001))
            s = int(off) + p + 100000
            mq[s:s + L] //= 4
    return depth, mq


def make_bedgraph_text(torch, dev, n, seed, mq, ctg_len=10_000_000):
    """n lines `ptg%06dl\t%09d\t%09d\t%02d\n` (34 bytes; %d reads the zero-padded numbers the same) on the device: contigs of ctg_len
    positions (the last one shorter), depth around 30 with a 20 kb dip every 400 kb (mq: a quarter of the depth in a 20 kb segment every 500 kb)"""
    gpos = torch.arange(n, device=dev, dtype=torch.int64)
    ctg = gpos // ctg_len + 1
    pos = gpos - (ctg - 1) * ctg_len
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    depth = 28 + torch.randint(0, 5, (n,), device=dev, generator=g)
    depth = torch.where((pos % 400000) < 20000, depth // 5, depth)
    if mq:
        depth = torch.where(((pos + 100000) % 500000) < 20000, depth // 4, depth)
    out = torch.empty((n, 34), dtype=torch.uint8, device=dev)
    out[:, :10] = torch.tensor(list(b"ptg000000l"), dtype=torch.uint8, device=dev)
    out[:, 10] = 9
    out[:, 20] = 9
    out[:, 30] = 9
    out[:, 33] = 10

    def digits(v, col, nd):
        for k in range(nd):
            out[:, col + nd - 1 - k] = (v % 10 + 48).to(torch.uint8)
            v = v // 10

    digits(ctg, 3, 6)
    digits(pos.clone(), 11, 9)
    digits(pos + 1, 21, 9)
    digits(depth.clone(), 31, 2)
    return out.reshape(-1)


FQ_HEAD = b"@read%07d runid=5c1f3b2a9d ch=%03d\n"


def make_fastq_piece(torch, dev, target_bases, seed):
    """One piece of ONT-like FASTQ text built on the device: read lengths log-normal(mu 9.2, sigma 0.9)
    clipped to [200, 200 000], uniform bases with a 200-base poly-A / poly-T stretch in every fourth read, qualities U[3, 40] + 33,
    header `@read%07d runid=... ch=%03d`.  -> (uint8 tensor of the text, read lengths, byte offset of every record)"""
    rng = np.random.default_rng(seed)
    m = int(target_bases / 14000 * 1.3) + 16
    L = np.clip(rng.lognormal(9.2, 0.9, size=m), 200, 200000).astype(np.int64)
    n = int(np.searchsorted(np.cumsum(L), target_bases)) + 1
    L = L[:n]
    hl = len(FQ_HEAD % (0, 0))
    size = hl + 2 * L + 4
    off = np.concatenate([[0], np.cumsum(size)[:-1]]).astype(np.int64)
    total = int(size.sum())
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    text = torch.empty(total, dtype=torch.uint8, device=dev)
    step = 1 << 28
    for s0 in range(0, total, step):
        e0 = min(total, s0 + step)
        text[s0:e0] = lut[torch.randint(0, 4, (e0 - s0,), device=dev, generator=g)]
    # quality bytes: [hl + L + 3, hl + 2 L + 3) of every record
    seg = np.empty(2 * n, dtype=np.int64)
    seg[0::2] = hl + L + 3
    seg[1::2] = L + 1
    isq = torch.repeat_interleave(torch.tensor([0, 1], dtype=torch.uint8, device=dev).repeat(n), torch.from_numpy(seg).to(dev)).bool()
    for s0 in range(0, total, step):
        e0 = min(total, s0 + step)
        q = torch.randint(36, 74, (e0 - s0,), dtype=torch.uint8, device=dev, generator=g)
        text[s0:e0] = torch.where(isq[s0:e0], q, text[s0:e0])
    del isq
    offd = torch.from_numpy(off).to(dev)
    Ld = torch.from_numpy(L).to(dev)
    heads = np.frombuffer(b"".join(FQ_HEAD % (i, i % 512) for i in range(n)), dtype=np.uint8).reshape(n, hl)
    text[(offd[:, None] + torch.arange(hl, device=dev)[None, :]).reshape(-1)] = torch.from_numpy(heads.copy()).to(dev).reshape(-1)
    sep = offd + hl + Ld
    text[sep] = 10
    text[sep + 1] = 43
    text[sep + 2] = 10
    text[offd + torch.from_numpy(size).to(dev) - 1] = 10
    pick = np.nonzero((np.arange(n) % 4 == 0) & (L > 400))[0]
    if len(pick):
        pa = (offd[torch.from_numpy(pick).to(dev)] + hl + 100)[:, None] + torch.arange(200, device=dev)[None, :]
        letter = torch.where(torch.from_numpy((pick % 8 == 0)).to(dev), torch.tensor(84, dtype=torch.uint8, device=dev), torch.tensor(65, dtype=torch.uint8, device=dev))
        text[pa.reshape(-1)] = letter[:, None].expand(-1, 200).reshape(-1)
    return text, L, off
