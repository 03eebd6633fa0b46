"""Reference CBOR image of MachineProof — TEST INFRASTRUCTURE, independent of valida_amd/csrc/host/cbor.hpp.

Builds the serde data model of the proof (nested dicts / lists, machine/src/proof.rs:13-44 + SURVEY.md Appendix B12)
from the VPF1 words, encodes it with a generic CBOR encoder (definite lengths, shortest-form integers — what ciborium
emits), and decodes CBOR back for round-trip checks.  The out-of-tree field names and the BabyBear / digest encodings are
UNPINNED (the Plonky3 fork is absent): the same two switches as the product encoder."""
import struct

P = 0x78000001
R = (1 << 32) % P


def model(words, canonical_fields=False, plain_digests=False):
    w = [int(x) for x in words]
    pos = [0]

    def take(k=1):
        v = w[pos[0]:pos[0] + k]
        assert len(v) == k, "truncated"
        pos[0] += k
        return v if k > 1 else v[0]

    def val(c):
        return c if canonical_fields else {"value": c * R % P}

    def ext(e):
        a = [val(x) for x in e]
        return a if canonical_fields else {"value": a}

    def digest(d):
        a = [val(x) for x in d]
        return a if plain_digests else {"value": a, "_marker": None}

    def path():
        return [[val(x) for x in take(8)] for _ in range(take())]

    assert take() == 0x31465056
    nc = take()
    roots = [take(8) for _ in range(3)]
    chips = []
    for _ in range(nc):
        cp = {"log_degree": take()}
        vecs = [[ext(take(5)) for _ in range(take())] for _ in range(5)]
        cp["opened_values"] = {
            "preprocessed_local": [], "preprocessed_next": [], "trace_local": vecs[0], "trace_next": vecs[1],
            "permutation_local": vecs[2], "permutation_next": vecs[3], "quotient_chunks": vecs[4],
        }
        cp["cumulative_sum"] = ext(take(5))
        chips.append(cp)
    fri = {"commit_phase_commits": [digest(take(8)) for _ in range(take())]}
    fri["query_proofs"] = [
        {"commit_phase_openings": [{"sibling_value": ext(take(5)), "opening_proof": path()} for _ in range(take())]} for _ in range(take())
    ]
    fri["final_poly"] = ext(take(5))
    fri["pow_witness"] = val(take())
    qo = []
    for _ in range(take()):
        rounds = []
        for _ in range(take()):
            nm = take()
            ov = []
            for _ in range(nm):
                wd = take()
                ov.append([val(take()) for _ in range(wd)])
            rounds.append({"opened_values": ov, "opening_proof": path()})
        qo.append(rounds)
    assert pos[0] == len(w), "trailing words"
    return {
        "commitments": {"main_trace": digest(roots[0]), "perm_trace": digest(roots[1]), "quotient_chunks": digest(roots[2])},
        "opening_proof": {"fri_proof": fri, "query_openings": qo},
        "chip_proofs": chips,
    }


def _head(major, v):
    m = major << 5
    if v < 24:
        return bytes([m | v])
    if v <= 0xFF:
        return bytes([m | 24, v])
    if v <= 0xFFFF:
        return bytes([m | 25]) + struct.pack(">H", v)
    if v <= 0xFFFFFFFF:
        return bytes([m | 26]) + struct.pack(">I", v)
    return bytes([m | 27]) + struct.pack(">Q", v)


def encode(x):
    if x is None:
        return b"\xf6"
    if isinstance(x, int):
        return _head(0, x)
    if isinstance(x, str):
        b = x.encode()
        return _head(3, len(b)) + b
    if isinstance(x, list):
        return _head(4, len(x)) + b"".join(encode(e) for e in x)
    if isinstance(x, dict):  # insertion order = struct declaration order
        return _head(5, len(x)) + b"".join(encode(k) + encode(v) for k, v in x.items())
    raise TypeError(type(x))


def decode(b):
    def item(i):
        ib = b[i]
        major, info = ib >> 5, ib & 31
        i += 1
        if ib == 0xF6:
            return None, i
        if info < 24:
            v = info
        else:
            n = {24: 1, 25: 2, 26: 4, 27: 8}[info]
            v = int.from_bytes(b[i:i + n], "big")
            i += n
        if major == 0:
            return v, i
        if major == 3:
            return b[i:i + v].decode(), i + v
        if major == 4:
            out = []
            for _ in range(v):
                e, i = item(i)
                out.append(e)
            return out, i
        if major == 5:
            out = {}
            for _ in range(v):
                k, i = item(i)
                out[k], i = item(i)
            return out, i
        raise ValueError("unsupported major type %d" % major)

    v, i = item(0)
    assert i == len(b), "trailing bytes"
    return v


def words_from_model(m):
    """Inverse of model(): the VPF1 words of a decoded CBOR proof (either variant of the two switches)."""
    rinv = pow(R, -1, P)

    def val(v):
        return v["value"] * rinv % P if isinstance(v, dict) else v

    def ext(e):
        return [val(x) for x in (e["value"] if isinstance(e, dict) else e)]

    def digest(d):
        return [val(x) for x in (d["value"] if isinstance(d, dict) else d)]

    def path(p):
        out = [len(p)]
        for node in p:
            out += [val(x) for x in node]
        return out

    w = [0x31465056, len(m["chip_proofs"])]
    for k in ("main_trace", "perm_trace", "quotient_chunks"):
        w += digest(m["commitments"][k])
    for cp in m["chip_proofs"]:
        ov = cp["opened_values"]
        assert ov["preprocessed_local"] == [] and ov["preprocessed_next"] == []
        w.append(cp["log_degree"])
        for k in ("trace_local", "trace_next", "permutation_local", "permutation_next", "quotient_chunks"):
            w.append(len(ov[k]))
            for e in ov[k]:
                w += ext(e)
        w += ext(cp["cumulative_sum"])
    fri = m["opening_proof"]["fri_proof"]
    w.append(len(fri["commit_phase_commits"]))
    for c in fri["commit_phase_commits"]:
        w += digest(c)
    w.append(len(fri["query_proofs"]))
    for q in fri["query_proofs"]:
        w.append(len(q["commit_phase_openings"]))
        for step in q["commit_phase_openings"]:
            w += ext(step["sibling_value"]) + path(step["opening_proof"])
    w += ext(fri["final_poly"]) + [val(fri["pow_witness"])]
    qo = m["opening_proof"]["query_openings"]
    w.append(len(qo))
    for rounds in qo:
        w.append(len(rounds))
        for bo in rounds:
            w.append(len(bo["opened_values"]))
            for row in bo["opened_values"]:
                w += [len(row)] + [val(x) for x in row]
            w += path(bo["opening_proof"])
    return w
