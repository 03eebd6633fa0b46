"""Keccak-f[1600] for gfx950 as ONE hand-scheduled inline-asm block with fixed VGPRs.

Why: tools/issue_patterns.py measured that a half-rate VALU instruction (v_alignbit_b32) directly followed by another VALU instruction of the
same wave makes BOTH cost a full 4-cycle SIMD slot, and that one scalar no-op behind every half-rate instruction restores the additive
cost (2.3 + 2.3 + 4.15 cycles for bitop3, bitop3, alignbit).  hipcc cannot be told to schedule that way, so the permutation's instruction
stream is generated here: a list scheduler over the real dependency graph (RAW / WAR / WAW on the fixed registers) that emits the repeating
slot pattern  full, full, half + s_nop  wherever the graph allows it.

Register plan (v0..v7 are left to the compiler):
  A (state)  word w = 2 * lane + half  ->  v[8 + w]          (v8..v57; bound to four 16-register operand tuples v[8:23] .. v[56:71])
  B (rho/pi) word w                   ->  v[58 + w]          (v58..v107)
  C (column parity) 2 * x + half      ->  v[108 + ..]        (v108..v117)
  R (C rotated by 1)                  ->  v[118 + ..]        (v118..v127)
"""
import sys

ROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001, 0x8000000080008081,
      0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B,
      0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A, 0x8000000080008081,
      0x8000000000008080, 0x0000000080000001, 0x8000000080008008]

A0, B0, C0, R0 = 8, 58, 108, 118
LANE_ORDER = "dest_row"  # or "source" (x fastest within y, the textbook order)


def A(lane, half): return A0 + 2 * lane + half
def B(lane, half): return B0 + 2 * lane + half
def C(x, half): return C0 + 2 * x + half
def R(x, half): return R0 + 2 * x + half


PHYS = {r: r for r in range(8, 128)}  # logical register -> physical VGPR (bank_optimise() permutes it)


class Ins:
    __slots__ = ("cls", "fmt", "dst", "srcs", "deps", "idx")

    def __init__(self, cls, fmt, dst, srcs):
        self.cls, self.fmt, self.dst, self.srcs = cls, fmt, dst, srcs

    @property
    def text(self):
        return self.fmt.format(d=PHYS[self.dst], s=[PHYS[x] for x in self.srcs])


def xor3(d, a, b, c): return Ins("B", "v_bitop3_b32 v{d}, v{s[0]}, v{s[1]}, v{s[2]} bitop3:0x96", d, (a, b, c))
def chi(d, a, b, c): return Ins("B", "v_bitop3_b32 v{d}, v{s[0]}, v{s[1]}, v{s[2]} bitop3:0xd2", d, (a, b, c))
def align(d, hi, lo, s): return Ins("A", "v_alignbit_b32 v{d}, v{s[0]}, v{s[1]}, %d" % s, d, (hi, lo))
def mov(d, a): return Ins("B", "v_mov_b32 v{d}, v{s[0]}", d, (a,))
def xorlit(d, lit): return Ins("B", "v_xor_b32 v{d}, 0x%x, v{s[0]}" % lit, d, (d,))


def parity_conflicts(ins):
    """number of instructions whose THREE register sources all have the same parity — the placement that costs a 3-source instruction a
    half-rate slot on gfx950 when its neighbours do the same (profiles/r03_microbench_issue.txt); two-source instructions never conflict"""
    n = 0
    for i in ins:
        if len(i.srcs) == 3:
            n += len({PHYS[s] & 1 for s in i.srcs}) == 1
    return n


def bank_optimise(ins, seed=1, steps=60000, hi=127, model="mod4"):
    """permute PHYS (A words stay inside v8..v57, the operand tuples; the rest inside v58..v127) to minimise, model "mod4": same-bank source
    pairs with bank = register mod 4 (round 2's guess); model "parity": 3-source instructions with all sources of one parity (round 3's
    measurement), ties broken by fewer such instructions directly behind one another"""
    import random
    rnd = random.Random(seed)
    groups = [list(range(8, 58)), list(range(58, hi + 1))]
    uses = {}
    for k, i in enumerate(ins):
        for r in set(i.srcs):
            uses.setdefault(r, []).append(k)

    def bad(k):
        return len(ins[k].srcs) == 3 and len({PHYS[r] & 1 for r in ins[k].srcs}) == 1

    def cost_of(k):
        if model == "parity":
            c = 4 * bad(k)
            if c and k + 1 < len(ins) and bad(k + 1):
                c += 1
            if c and k and bad(k - 1):
                c += 1
            return c
        b = [PHYS[r] % 4 for r in set(ins[k].srcs)]
        return len(b) - len(set(b))

    def same(a, b):
        return (PHYS[a] & 1) == (PHYS[b] & 1) if model == "parity" else PHYS[a] % 4 == PHYS[b] % 4

    total = sum(cost_of(k) for k in range(len(ins)))
    for _ in range(steps):
        g = groups[rnd.random() < 0.6]
        a, b = rnd.sample(g, 2)
        if same(a, b):
            continue
        touched = set(uses.get(a, [])) | set(uses.get(b, []))
        if model == "parity":
            touched |= {k + d for k in touched for d in (-1, 1) if 0 <= k + d < len(ins)}
        before = sum(cost_of(k) for k in touched)
        PHYS[a], PHYS[b] = PHYS[b], PHYS[a]
        after = sum(cost_of(k) for k in touched)
        if after > before:
            PHYS[a], PHYS[b] = PHYS[b], PHYS[a]
        else:
            total += after - before
    return total


def rotl_pair(lo, hi, olo, ohi, n):
    """(olo, ohi) = rotl64((lo, hi), n) as instructions"""
    if n == 0:
        return [mov(olo, lo), mov(ohi, hi)]
    if n == 32:
        return [mov(olo, hi), mov(ohi, lo)]
    if n < 32:
        return [align(ohi, hi, lo, 32 - n), align(olo, lo, hi, 32 - n)]
    return [align(ohi, lo, hi, 64 - n), align(olo, hi, lo, 64 - n)]


def round_instrs(rnd, digest_only_last=False):
    out = []
    for x in range(5):
        for h in (0, 1):
            out.append(xor3(C(x, h), A(x, h), A(x + 5, h), A(x + 10, h)))
            out.append(xor3(C(x, h), C(x, h), A(x + 15, h), A(x + 20, h)))
    for x in range(5):
        out += rotl_pair(C(x, 0), C(x, 1), R(x, 0), R(x, 1), 1)
    lanes = [(x, y) for y in range(5) for x in range(5)]
    if digest_only_last:
        lanes = [(x, x) for x in range(5)]  # row 0 of the output reads B[0..4], whose pi-preimages are the diagonal lanes
    # order the lanes by DESTINATION row so that chi of a row can start while the next row's lanes are still being rotated
    if LANE_ORDER == "dest_row":
        lanes.sort(key=lambda xy: ((2 * xy[0] + 3 * xy[1]) % 5, xy[1]))
    for x, y in lanes:
        src, dst = x + 5 * y, y + 5 * ((2 * x + 3 * y) % 5)
        for h in (0, 1):
            out.append(xor3(A(src, h), A(src, h), C((x + 4) % 5, h), R((x + 1) % 5, h)))  # theta applied in place
        if ROT[src] == 0:  # lane (0,0): no rotation and pi maps it to itself: chi reads it where it is (b(0) below)
            pass
        else:
            out += rotl_pair(A(src, 0), A(src, 1), B(dst, 0), B(dst, 1), ROT[src])
    b = lambda lane, h: A(0, h) if lane == 0 else B(lane, h)
    rows = range(1) if digest_only_last else range(5)
    for y in rows:
        # row 0 writes lane 0 LAST: the other outputs of the row still read the unrotated lane 0 from A(0)
        xs = ([1, 2, 3, 0] if digest_only_last else [1, 2, 3, 4, 0]) if y == 0 else list(range(5))
        for x in xs:
            for h in (0, 1):
                out.append(chi(A(x + 5 * y, h), b(x + 5 * y, h), b((x + 1) % 5 + 5 * y, h), b((x + 2) % 5 + 5 * y, h)))
    lo, hi = RC[rnd] & 0xffffffff, RC[rnd] >> 32
    if lo:
        out.append(xorlit(A(0, 0), lo))
    if hi:
        out.append(xorlit(A(0, 1), hi))
    return out


# ---- in-place variant: 80 registers.  B is only ever one row (T, 5 lanes); chi of that row writes into the register pairs its five source lanes
# just vacated, so a lane's home moves by pi^-1 every round -- and pi has order 24 on the 24 off-origin lanes, so after 24 rounds every lane is home.
T0I, C0I, R0I = 58, 68, 78  # T v58..v67, C v68..v77, R v78..v87


def inplace_permutation(rounds=24, digest_only=False, zero_lanes=()):
    """-> (instructions, loc) ; loc[lane] = register-pair index (A0 + 2 * pair + half) where the lane ends up (identity after 24 full rounds)"""
    loc = list(range(25))
    Areg = lambda pair, h: A0 + 2 * pair + h
    Treg = lambda x, h: T0I + 2 * x + h
    Creg = lambda x, h: C0I + 2 * x + h
    Rreg = lambda x, h: R0I + 2 * x + h
    out = []
    for rnd in range(rounds):
        last_digest = digest_only and rnd == rounds - 1
        for x in range(5):
            for h in (0, 1):
                out.append(xor3(Creg(x, h), Areg(loc[x], h), Areg(loc[x + 5], h), Areg(loc[x + 10], h)))
                out.append(xor3(Creg(x, h), Creg(x, h), Areg(loc[x + 15], h), Areg(loc[x + 20], h)))
        for x in range(5):
            out += rotl_pair(Creg(x, 0), Creg(x, 1), Rreg(x, 0), Rreg(x, 1), 1)
        new_loc = list(loc)
        for yd in (range(1) if last_digest else range(5)):
            srcs = []
            for xd in range(5):
                # source lane (x, y) with pi(x, y) = (xd, yd): xd = y, yd = (2x + 3y) % 5
                y = xd
                x = next(x for x in range(5) if (2 * x + 3 * y) % 5 == yd)
                src = x + 5 * y
                srcs.append(src)
                pr = loc[src]
                for h in (0, 1):
                    out.append(xor3(Areg(pr, h), Areg(pr, h), Creg((x + 4) % 5, h), Rreg((x + 1) % 5, h)))
                if ROT[src]:
                    out += rotl_pair(Areg(pr, 0), Areg(pr, 1), Treg(xd, 0), Treg(xd, 1), ROT[src])
            b = lambda xd, h: Areg(loc[srcs[xd]], h) if ROT[srcs[xd]] == 0 else Treg(xd, h)
            # an output may only overwrite a pair whose unrotated value nobody still reads: lane 0 (never rotated) is read from its home as b(0)
            xs = list(range(4 if last_digest else 5))
            if ROT[srcs[0]] == 0:
                xs = xs[1:] + xs[:1]
            for xd in xs:
                pr = loc[srcs[xd]]
                for h in (0, 1):
                    out.append(chi(Areg(pr, h), b(xd, h), b((xd + 1) % 5, h), b((xd + 2) % 5, h)))
                new_loc[xd + 5 * yd] = pr
        loc = new_loc
        lo, hi = RC[rnd] & 0xffffffff, RC[rnd] >> 32
        if lo:
            out.append(xorlit(Areg(loc[0], 0), lo))
        if hi:
            out.append(xorlit(Areg(loc[0], 1), hi))
    return out, loc


def simulate_inplace(lines, state, loc):
    v = _run(lines, {PHYS[A(l, 0)]: state[l] & 0xffffffff for l in range(25)} | {PHYS[A(l, 1)]: state[l] >> 32 for l in range(25)})
    return [v[PHYS[A0 + 2 * loc[l]]] | (v[PHYS[A0 + 2 * loc[l] + 1]] << 32) for l in range(25)]


def add_deps(ins):
    last_write, readers = {}, {}
    for i, n in enumerate(ins):
        n.idx = i
        deps = set()
        for s in n.srcs:
            if s in last_write:
                deps.add(last_write[s])
        if n.dst in last_write:
            deps.add(last_write[n.dst])
        for r in readers.get(n.dst, ()):
            if r != i:
                deps.add(r)
        n.deps = deps
        for s in n.srcs:
            readers.setdefault(s, []).append(i)
        last_write[n.dst] = i
        readers[n.dst] = [i] if n.dst in n.srcs else []
    return ins


def with_nops(ins_seq, mode):
    """asm lines of an instruction sequence; mode: none | each (s_nop 0 behind every half-rate instruction) | pair (behind a RUN of half-rate
    instructions) | before (in front of a run) | each1 (s_nop 1 behind every one)"""
    lines = []
    for k, i in enumerate(ins_seq):
        nxt = ins_seq[k + 1].cls if k + 1 < len(ins_seq) else "B"
        prv = ins_seq[k - 1].cls if k else "B"
        if mode == "before" and i.cls == "A" and prv != "A":
            lines.append("s_nop 0")
        lines.append(i.text)
        if i.cls == "A":
            if mode == "each":
                lines.append("s_nop 0")
            elif mode == "each1":
                lines.append("s_nop 1")
            elif mode == "pair" and nxt != "A":
                lines.append("s_nop 0")
    return lines


def bank_conflicts(ins):
    """number of instructions with two or more VGPR sources in the same bank (register number mod 4)"""
    n = 0
    for i in ins:
        banks = [PHYS[s] % 4 for s in set(i.srcs)]
        n += len(banks) != len(set(banks))
    return n


def schedule(ins, pattern="BBA", min_dist=1, window=400, nop_after_half=True):
    """List-schedule `ins` (program order = priority) into the repeating class pattern; returns asm lines."""
    n = len(ins)
    emitted_at = [None] * n
    done = 0
    pending = list(range(n))
    lines, slot, pos = [], 0, 0
    while pending:
        want = pattern[slot % len(pattern)]
        pick = None
        fallback = None
        for k in pending[:window]:
            i = ins[k]
            if all(emitted_at[d] is not None and pos - emitted_at[d] >= min_dist for d in i.deps):
                if i.cls == want:
                    pick = k
                    break
                if fallback is None:
                    fallback = k
        if pick is None:
            pick = fallback
        if pick is None:  # nothing satisfies the distance: take the oldest whose producers are at least emitted
            for k in pending[:window]:
                if all(emitted_at[d] is not None for d in ins[k].deps):
                    pick = k
                    break
        i = ins[pick]
        pending.remove(pick)
        emitted_at[pick] = pos
        pos += 1
        lines.append(i.text)
        if i.cls == "A" and nop_after_half:
            lines.append("s_nop 0")
        slot += 1
    return lines


def permutation(digest_only=False, pattern="BBA", min_dist=1, nop=True, rounds=24):
    ins = []
    for r in range(rounds):
        ins += round_instrs(r, digest_only and r == rounds - 1)
    add_deps(ins)
    return schedule(ins, pattern, min_dist, nop_after_half=nop), ins


def reference(state, rounds=24):
    """plain Keccak-p on 25 64-bit lanes (the generator's own check of its instruction list is the GPU comparison)"""
    a = list(state)
    M = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    for r in range(rounds):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        b = [0] * 25
        for y in range(5):
            for x in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y] ^ d[x], ROT[x + 5 * y])
        a = [b[i] ^ (~b[(i % 5 + 1) % 5 + 5 * (i // 5)] & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) & M for i in range(25)]
        a[0] ^= RC[r]
    return a


def simulate(lines, state):
    """execute the generated asm text on a Python register file (checks scheduling + register allocation without a GPU)"""
    v = {}
    for lane in range(25):
        v[PHYS[A(lane, 0)]] = state[lane] & 0xffffffff
        v[PHYS[A(lane, 1)]] = state[lane] >> 32
    v = _run(lines, v)
    return [v[PHYS[A(lane, 0)]] | (v[PHYS[A(lane, 1)]] << 32) for lane in range(25)]


def _run(lines, v):
    for ln in lines:
        t = ln.replace(",", " ").split()
        if t[0] == "s_nop":
            continue
        d = int(t[1][1:])
        if t[0] == "v_bitop3_b32":
            a, b, c = (v[int(x[1:])] for x in t[2:5])
            tt = int(t[5].split(":")[1], 16)
            r = 0
            for bit in range(32):
                idx = (((a >> bit) & 1) << 2) | (((b >> bit) & 1) << 1) | ((c >> bit) & 1)
                r |= ((tt >> idx) & 1) << bit
            v[d] = r
        elif t[0] == "v_alignbit_b32":
            hi, lo, s = v[int(t[2][1:])], v[int(t[3][1:])], int(t[4])
            v[d] = (((hi << 32) | lo) >> s) & 0xffffffff
        elif t[0] == "v_mov_b32":
            v[d] = v[int(t[2][1:])]
        elif t[0] == "v_xor_b32":
            v[d] = int(t[2], 16) ^ v[int(t[3][1:])]
        else:
            raise ValueError(ln)
    return v


def emit_header(path):
    """valida_amd/csrc/kernels/keccak_f1600_gfx950.inc: the permutation as asm string macros (identity register map, program order,
    s_nop 0 behind every v_alignbit_b32).  State word w (= 2 * lane + half) is v[8 + w] on entry and on exit."""
    PHYS.update({r: r for r in range(8, 128)})
    out = ["// GENERATED by tools/keccak_asm.py --emit -- do not edit (tests/test_host_cpu.py regenerates and compares).",
           "// Keccak-f[1600] for gfx950, one permutation per lane, fixed VGPRs: state word w = 2 * lane + half in v[8 + w] (v8..v57), scratch v58..v127.",
           "// Bind the state with the operand tuples \"+{v[8:23]}\", \"+{v[24:39]}\", \"+{v[40:55]}\", \"+{v[56:71]}\" and clobber VK_KECCAK_ASM_CLOBBERS."]
    for name, digest in (("VK_KECCAK_F1600_ASM", False), ("VK_KECCAK_F1600_DIGEST_ASM", True)):
        ins = []
        for r in range(24):
            ins += round_instrs(r, digest and r == 23)
        lines = with_nops(ins, "each")
        out.append("#define %s \\" % name)
        out += ['    "%s\\n" \\' % ln for ln in lines[:-1]]
        out.append('    "%s\\n"' % lines[-1])
    out.append("#define VK_KECCAK_ASM_CLOBBERS " + ", ".join('"v%d"' % r for r in range(72, 128)))
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")


if __name__ == "__main__":
    import os
    if len(sys.argv) > 1 and sys.argv[1] == "--emit":
        emit_header(sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "valida_amd", "csrc", "kernels",
                                                                      "keccak_f1600_gfx950.inc"))
        sys.exit(0)
    import random
    rnd = random.Random(1)
    st = [rnd.getrandbits(64) for _ in range(25)]
    for pat, dist in (("BBA", 1), ("BBA", 3)):
        lines, ins = permutation(False, pat, dist)
        got = simulate(lines, st)
        assert got == reference(st), "schedule %s/%d is wrong" % (pat, dist)
        nB = sum(1 for i in ins if i.cls == "B")
        nA = sum(1 for i in ins if i.cls == "A")
        print(pat, dist, "ok:", len(ins), "instructions", nB, "full", nA, "half")
    lines, ins = permutation(True)
    got = simulate(lines, st)
    ref = reference(st)
    assert got[:4] == ref[:4]
    print("digest-only ok:", len(ins))
    one = add_deps(round_instrs(1))
    print("same-bank source pairs in one round:", bank_conflicts(one), "of", len(one), "instructions")
    print("after bank_optimise:", bank_optimise(one), bank_conflicts(one))
    lines, ins = permutation(False)
    assert simulate(lines, st) == reference(st)
    print("remapped registers ok")
    PHYS.update({r: r for r in range(8, 128)})
    ins, loc = inplace_permutation()
    assert loc == list(range(25)), loc
    lines = with_nops(ins, "each")
    assert simulate_inplace(lines, st, loc) == reference(st)
    used = {r for i in ins for r in (i.dst,) + tuple(i.srcs)}
    print("in-place ok:", len(ins), "instructions, registers v%d..v%d" % (min(used), max(used)))
    ins, loc = inplace_permutation(digest_only=True)
    got = simulate_inplace(with_nops(ins, "each"), st, loc)
    assert got[:4] == reference(st)[:4]
    print("in-place digest-only ok:", len(ins), "digest lanes at pairs", loc[:4])
