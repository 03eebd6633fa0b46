"""ORACLE — TEST INFRASTRUCTURE ONLY.  ctypes loader for oracle/liboracle.so (the CPU restatement).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (valida_amd/) never does.  PARITY UNPINNED vs Plonky3@bdd338d6 (see field.hpp).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboracle.so")
c_u32p = ctypes.POINTER(ctypes.c_uint32)
_lib = None


def usable_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (a container that shows 256 CPUs
    but is throttled to a few would otherwise run 256 spinning OpenMP threads on them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, quota // int(g.read())))
        except (OSError, ValueError, IndexError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise ImportError("oracle/liboracle.so is not built: run `make -C oracle`")
        # before libgomp initialises: no more threads than usable cores, and sleeping (not spinning) waits
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = ctypes.CDLL(_PATH)
        L.oracle_two_adic_generator.restype = ctypes.c_uint32
        L.oracle_fp_mul.restype = ctypes.c_uint32
        L.oracle_fp_inv.restype = ctypes.c_uint32
        L.oracle_grind.restype = ctypes.c_uint32
        L.oracle_log_quotient_degree.restype = ctypes.c_uint32
        L.oracle_num_interactions.restype = ctypes.c_uint32
        L.oracle_prove_basic.restype = ctypes.c_void_p
        L.oracle_result_len.restype = ctypes.c_uint64
        L.oracle_result_words.restype = c_u32p
        L.oracle_result_seconds.restype = ctypes.c_double
        L.oracle_result_perm_trace.restype = ctypes.c_uint64
        L.oracle_result_quotient.restype = ctypes.c_uint64
        _lib = L
    return _lib


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(c_u32p)


def keccak256(data: bytes, pad=0x01) -> bytes:
    out = (ctypes.c_uint8 * 32)()
    buf = (ctypes.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b"\0")
    lib().oracle_keccak256(buf, ctypes.c_uint64(len(data)), out, ctypes.c_uint32(pad))
    return bytes(out)


def keccak_f1600(state):
    s = np.array(state, dtype=np.uint64)
    lib().oracle_keccak_f1600(s.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    return s


def hash_elems(e):
    a, ap = _u32(e)
    out = np.zeros(8, dtype=np.uint32)
    lib().oracle_hash_elems(ap, ctypes.c_uint64(a.size), out.ctypes.data_as(c_u32p))
    return out


def compress(a, b):
    x, xp = _u32(a)
    y, yp = _u32(b)
    out = np.zeros(8, dtype=np.uint32)
    lib().oracle_compress(xp, yp, out.ctypes.data_as(c_u32p))
    return out


def poseidon_permute(rc, state):
    r, rp = _u32(rc)
    s = np.array(state, dtype=np.uint32)
    lib().oracle_poseidon_permute(rp, s.ctypes.data_as(c_u32p))
    return s


def challenger_probe(rc, observed, n_samples):
    r, rp = _u32(rc)
    o, op = _u32(observed)
    out = np.zeros(n_samples, dtype=np.uint32)
    lib().oracle_challenger_probe(rp, op, ctypes.c_uint64(o.size), out.ctypes.data_as(c_u32p), ctypes.c_uint64(n_samples))
    return out


def grind(rc, observed, bits):
    r, rp = _u32(rc)
    o, op = _u32(observed)
    return int(lib().oracle_grind(rp, op, ctypes.c_uint64(o.size), ctypes.c_uint32(bits)))


def dft(col, inverse=False):
    a = np.array(col, dtype=np.uint32)
    lib().oracle_dft(a.ctypes.data_as(c_u32p), ctypes.c_uint64(a.size), ctypes.c_int(1 if inverse else 0))
    return a


def naive_dft(col):
    a, ap = _u32(col)
    out = np.zeros_like(a)
    lib().oracle_naive_dft(ap, ctypes.c_uint64(a.size), out.ctypes.data_as(c_u32p))
    return out


def committed_lde(m, added_bits=1, lde_shift=31):
    a, ap = _u32(m)
    h, w = a.shape
    out = np.zeros((h << added_bits, w), dtype=np.uint32)
    lib().oracle_committed_lde(ap, ctypes.c_uint64(h), ctypes.c_uint64(w), ctypes.c_uint32(added_bits), ctypes.c_uint32(lde_shift), out.ctypes.data_as(c_u32p))
    return out


def _mat_args(mats):
    keep = [np.ascontiguousarray(m, dtype=np.uint32) for m in mats]
    ptrs = (c_u32p * len(keep))(*[k.ctypes.data_as(c_u32p) for k in keep])
    hs = (ctypes.c_uint64 * len(keep))(*[k.shape[0] for k in keep])
    ws = (ctypes.c_uint64 * len(keep))(*[k.shape[1] for k in keep])
    return keep, ptrs, hs, ws


def commit_root(mats, shifts=None, log_blowup=1):
    keep, ptrs, hs, ws = _mat_args(mats)
    out = np.zeros(8, dtype=np.uint32)
    sp = None
    if shifts is not None:
        sv, sp = _u32(shifts)
    lib().oracle_commit_root(ctypes.c_uint64(len(keep)), ptrs, hs, ws, sp, ctypes.c_uint32(log_blowup), out.ctypes.data_as(c_u32p))
    return out


def mmcs_root(mats):
    keep, ptrs, hs, ws = _mat_args(mats)
    out = np.zeros(8, dtype=np.uint32)
    lib().oracle_mmcs_root(ctypes.c_uint64(len(keep)), ptrs, hs, ws, out.ctypes.data_as(c_u32p))
    return out


def perm_trace(chip, main, challenges15):
    m, mp = _u32(main)
    c, cp = _u32(challenges15)
    M = int(lib().oracle_num_interactions(ctypes.c_uint32(chip)))
    out = np.zeros((m.shape[0], 5 * (M + 1)), dtype=np.uint32)
    lib().oracle_perm_trace(ctypes.c_uint32(chip), mp, ctypes.c_uint64(m.shape[0]), cp, out.ctypes.data_as(c_u32p))
    return out


def eval_constraints(chip, local, nxt, prep_local=None, prep_next=None, is_first=0, is_last=0, is_transition=1):
    a, ap = _u32(local)
    b, bp = _u32(nxt)
    c, cp = _u32(prep_local if prep_local is not None else np.zeros(8, dtype=np.uint32))
    d, dp = _u32(prep_next if prep_next is not None else np.zeros(8, dtype=np.uint32))
    L = lib()
    L.oracle_eval_constraints.restype = ctypes.c_uint32
    n = L.oracle_eval_constraints(ctypes.c_uint32(chip), ap, bp, cp, dp, ctypes.c_uint32(is_first), ctypes.c_uint32(is_last), ctypes.c_uint32(is_transition), None,
                                  ctypes.c_uint32(0))
    out = np.zeros(max(1, n), dtype=np.uint32)
    L.oracle_eval_constraints(ctypes.c_uint32(chip), ap, bp, cp, dp, ctypes.c_uint32(is_first), ctypes.c_uint32(is_last), ctypes.c_uint32(is_transition),
                              out.ctypes.data_as(c_u32p), ctypes.c_uint32(out.size))
    return out[:n]


def interactions(chip):
    """all_interactions of chip `chip` from the oracle's own transcription, in the neutral word image (see capi.cpp)."""
    L = lib()
    L.oracle_interaction_words.restype = ctypes.c_uint32
    n = int(L.oracle_interaction_words(ctypes.c_uint32(chip), None, ctypes.c_uint32(0)))
    w = np.zeros(n, dtype=np.uint32)
    L.oracle_interaction_words(ctypes.c_uint32(chip), w.ctypes.data_as(c_u32p), ctypes.c_uint32(n))
    return w


def chip_shape(chip):
    out = np.zeros(2, dtype=np.uint32)
    lib().oracle_chip_shape(ctypes.c_uint32(chip), out.ctypes.data_as(c_u32p))
    return int(out[0]), int(out[1])


def fri_fold(f, beta5):
    a, ap = _u32(f)
    b, bp = _u32(beta5)
    out = np.zeros((a.shape[0] // 2, 5), dtype=np.uint32)
    lib().oracle_fri_fold(ap, ctypes.c_uint64(a.shape[0]), bp, out.ctypes.data_as(c_u32p))
    return out


def ext5_mul(a, b):
    x, xp = _u32(a)
    y, yp = _u32(b)
    out = np.zeros(5, dtype=np.uint32)
    lib().oracle_ext5_mul(xp, yp, out.ctypes.data_as(c_u32p))
    return out


def ext5_inv(a):
    x, xp = _u32(a)
    out = np.zeros(5, dtype=np.uint32)
    lib().oracle_ext5_inv(xp, out.ctypes.data_as(c_u32p))
    return out


class ProveResult:
    def __init__(self, handle):
        self._h = ctypes.c_void_p(handle)
        n = int(lib().oracle_result_len(self._h))
        self.words = np.ctypeslib.as_array(lib().oracle_result_words(self._h), shape=(n,)).copy()
        self.seconds = float(lib().oracle_result_seconds(self._h))
        t = np.zeros(33, dtype=np.uint32)
        lib().oracle_result_transcript(self._h, t.ctypes.data_as(c_u32p))
        self.transcript = t

    def bytes(self):
        return self.words.tobytes()

    def perm_trace(self, chip):
        n = int(lib().oracle_result_perm_trace(self._h, ctypes.c_uint32(chip), None, ctypes.c_uint64(0)))
        out = np.zeros(n, dtype=np.uint32)
        lib().oracle_result_perm_trace(self._h, ctypes.c_uint32(chip), out.ctypes.data_as(c_u32p), ctypes.c_uint64(n))
        return out

    def quotient(self, chip):
        n = int(lib().oracle_result_quotient(self._h, ctypes.c_uint32(chip), None, ctypes.c_uint64(0)))
        out = np.zeros(n, dtype=np.uint32)
        lib().oracle_result_quotient(self._h, ctypes.c_uint32(chip), out.ctypes.data_as(c_u32p), ctypes.c_uint64(n))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_result_free(self._h)
            self._h = None


def set_fast(on):
    """Fast mode (oracle/fast.hpp): identical proof words, computed the way a tuned CPU prover would (AVX2 Montgomery NTTs, unrolled Keccak,
    batch inversions).  Used by bench.py's cpu_baseline leg (kind "port-simd") and fixture generation; the scalar mode stays the checker."""
    lib().oracle_set_fast(ctypes.c_int(1 if on else 0))


def keep_heap():
    """oracle_keep_heap: freed memory stays in the heap (glibc mallopt) — for the one-proof baseline process only."""
    lib().oracle_keep_heap()


def set_mmcs_hash(kind, rc=None):
    """0: Keccak MMCS (the reference's); 1: Poseidon-16 sponge / truncated permutation with round constants `rc`."""
    if kind == 1:
        r, rp = _u32(rc)
        lib().oracle_set_mmcs_hash(ctypes.c_int(1), rp)
    else:
        lib().oracle_set_mmcs_hash(ctypes.c_int(0), None)


def pcs_open(rounds, points, rc, observed=(), log_blowup=1, num_queries=40, pow_bits=8):
    """pcs.commit_batches per round + pcs.open_multi_batches: rounds = [[matrix, ...], ...] (row-major), points[r][i] = list of
    Ext5 (5 words).  Returns (roots [n_rounds x 8], opened values flat, TwoAdicFriPcsProof words)."""
    L = lib()
    L.oracle_pcs_open.restype = ctypes.c_void_p
    L.oracle_pcs_open_len.restype = ctypes.c_uint64
    L.oracle_pcs_open_words.restype = c_u32p
    flat = [m for rnd in rounds for m in rnd]
    keep, ptrs, hs, ws = _mat_args(flat)
    n_mats = np.array([len(rnd) for rnd in rounds], dtype=np.uint32)
    n_points = np.array([len(pts) for rnd in points for pts in rnd], dtype=np.uint32)
    pw = np.array([w for rnd in points for pts in rnd for z in pts for w in z], dtype=np.uint32)
    r, rp = _u32(rc)
    o, op = _u32(np.array(observed, dtype=np.uint32))
    h = ctypes.c_void_p(L.oracle_pcs_open(ctypes.c_uint32(len(rounds)), n_mats.ctypes.data_as(c_u32p), ptrs, hs, ws, n_points.ctypes.data_as(c_u32p),
                                          pw.ctypes.data_as(c_u32p), rp, ctypes.c_uint32(log_blowup), ctypes.c_uint32(num_queries), ctypes.c_uint32(pow_bits), op,
                                          ctypes.c_uint64(o.size)))
    n = int(L.oracle_pcs_open_len(h))
    words = np.ctypeslib.as_array(L.oracle_pcs_open_words(h), shape=(n,)).copy()
    split = (ctypes.c_uint64 * 2)()
    L.oracle_pcs_open_split(h, split)
    L.oracle_pcs_open_free(h)
    a, b = int(split[0]), int(split[0]) + int(split[1])
    return words[:a].reshape(-1, 8), words[a:b], words[b:]


def set_observe_final_poly(on):
    """Convention switch of the restatement: absorb final_poly into the transcript before grinding."""
    lib().oracle_set_observe_final_poly(ctypes.c_int(1 if on else 0))


def prove_basic(main_traces, prep_program, prep_range, rc, log_blowup=1, num_queries=40, pow_bits=8, debug_check=False):
    keep = [np.ascontiguousarray(m, dtype=np.uint32) for m in main_traces]
    ptrs = (c_u32p * len(keep))(*[k.ctypes.data_as(c_u32p) for k in keep])
    hs = (ctypes.c_uint64 * len(keep))(*[k.shape[0] for k in keep])
    pp, ppp = _u32(prep_program)
    pr, prp = _u32(prep_range)
    r, rp = _u32(rc)
    h = lib().oracle_prove_basic(ptrs, hs, ppp, ctypes.c_uint64(pp.shape[0]), prp, rp, ctypes.c_uint32(log_blowup), ctypes.c_uint32(num_queries),
                                 ctypes.c_uint32(pow_bits), ctypes.c_int(1 if debug_check else 0))
    return ProveResult(h)


TEST_POW5, TEST_POW9 = 100, 101  # the oracle's synthetic higher-degree AIRs (chips.hpp)


def prove_machine(chip_ids, main_traces, rc, log_blowup=1, num_queries=40, pow_bits=8, debug_check=False):
    """Proof of a machine made of the given chip ids (no preprocessed traces): the general log_quotient_degree tests."""
    ids = np.array(chip_ids, dtype=np.uint32)
    keep = [np.ascontiguousarray(m, dtype=np.uint32) for m in main_traces]
    ptrs = (c_u32p * len(keep))(*[k.ctypes.data_as(c_u32p) for k in keep])
    hs = (ctypes.c_uint64 * len(keep))(*[k.shape[0] for k in keep])
    r, rp = _u32(rc)
    L = lib()
    L.oracle_prove_machine.restype = ctypes.c_void_p
    h = L.oracle_prove_machine(ids.ctypes.data_as(c_u32p), ctypes.c_uint32(ids.size), ptrs, hs, rp, ctypes.c_uint32(log_blowup), ctypes.c_uint32(num_queries),
                               ctypes.c_uint32(pow_bits), ctypes.c_int(1 if debug_check else 0))
    return ProveResult(h)


def verify_machine(chip_ids, proof_words, rc, log_blowup=1, num_queries=40, pow_bits=8):
    ids = np.array(chip_ids, dtype=np.uint32)
    w, wp = _u32(proof_words)
    r, rp = _u32(rc)
    msg = ctypes.create_string_buffer(256)
    rcode = lib().oracle_verify_machine(ids.ctypes.data_as(c_u32p), ctypes.c_uint32(ids.size), wp, ctypes.c_uint64(w.size), rp, ctypes.c_uint32(log_blowup),
                                        ctypes.c_uint32(num_queries), ctypes.c_uint32(pow_bits), msg, ctypes.c_uint64(256))
    return None if rcode == 0 else msg.value.decode()


def verify_basic(prep_program, prep_range, proof_words, rc, log_blowup=1, num_queries=40, pow_bits=8):
    """Returns None if the proof is accepted, else the rejection message."""
    pp, ppp = _u32(prep_program)
    pr, prp = _u32(prep_range)
    w, wp = _u32(proof_words)
    r, rp = _u32(rc)
    msg = ctypes.create_string_buffer(256)
    rcode = lib().oracle_verify_basic(ppp, ctypes.c_uint64(pp.shape[0]), prp, wp, ctypes.c_uint64(w.size), rp, ctypes.c_uint32(log_blowup),
                                      ctypes.c_uint32(num_queries), ctypes.c_uint32(pow_bits), msg, ctypes.c_uint64(256))
    return None if rcode == 0 else msg.value.decode()
