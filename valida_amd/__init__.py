"""valida_amd — MI355X-native STARK prover backend for the Valida zkVM (hot path: Machine::prove).

This package is a thin ctypes mirror of the C ABI in include/vgpu.h (libvgpu.so: hand-written
HIP/gfx950 kernels + C++ host prover).  Names follow the reference's own surface for this path:
`Machine.prove`, `Pcs.commit_batches` / `get_ldes`, `generate_permutation_trace`, `Challenger`.
There is NO CPU fallback: constructing a `Prover` without a HIP device raises `VgpuError`.
"""
import ctypes
import os

import numpy as np

P = 2013265921
NUM_CHIPS = 14
CHIP_NAMES = ["cpu", "program", "mem", "add", "sub", "mul", "div", "shift", "lt", "com", "bitwise", "output", "range", "static_data"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("VGPU_LIB_PATH") or os.path.join(_HERE, "libvgpu.so")  # the override serves A/B runs of two builds in one session

c_u32p = ctypes.POINTER(ctypes.c_uint32)
c_u64p = ctypes.POINTER(ctypes.c_uint64)


HASH_KECCAK256, HASH_POSEIDON16 = 0, 1  # vgpu_config.hash_kind


class VgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vgpu error %d: %s" % (code, msg))
        self.code = code


class VgpuConfig(ctypes.Structure):
    _fields_ = [
        ("device", ctypes.c_int32),
        ("log_blowup", ctypes.c_uint32),
        ("num_queries", ctypes.c_uint32),
        ("pow_bits", ctypes.c_uint32),
        ("hash_kind", ctypes.c_uint32),
        ("observe_final_poly", ctypes.c_uint32),
        ("poseidon_rc", ctypes.c_uint32 * 480),
        ("interpret_air", ctypes.c_uint32),
    ]


_lib = None


def lib():
    """Load libvgpu.so (built in-tree by valida_amd/build.py).  Fails loudly if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise ImportError("libvgpu.so is not built: run `python valida_amd/build.py` (or __graft_entry__.build())")
        L = ctypes.CDLL(_LIB_PATH)
        L.vgpu_last_error.restype = ctypes.c_char_p
        L.vgpu_version.restype = ctypes.c_char_p
        L.vgpu_proof_len.restype = ctypes.c_uint64
        L.vgpu_proof_words.restype = c_u32p
        L.vgpu_challenger_sample_bits.restype = ctypes.c_uint64
        L.vgpu_proof_debug_perm_trace.restype = ctypes.c_int64
        L.vgpu_proof_debug_quotient.restype = ctypes.c_int64
        L.vgpu_prover_profile.restype = ctypes.c_int64
        L.vgpu_shader_clock_probe.argtypes = [ctypes.c_int32, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]
        L.vgpu_shader_clock_probe.restype = ctypes.c_int32
        for name in ("vgpu_air_constant", "vgpu_air_variable", "vgpu_air_is_first_row", "vgpu_air_is_last_row", "vgpu_air_is_transition", "vgpu_air_add",
                     "vgpu_air_sub", "vgpu_air_mul", "vgpu_air_neg", "vgpu_machine_num_chips", "vgpu_challenger_grind"):
            getattr(L, name).restype = ctypes.c_uint32
        _lib = L
    return _lib


def _check(code):
    if code != 0:
        raise VgpuError(code, lib().vgpu_last_error().decode())


def _u32(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(c_u32p)


def shader_clock_hz(device=0, iters=8192):
    """The shader clock the device sustains right now (vgpu_shader_clock_probe: one wave, a chain of `iters` VALU additions, shader cycles
    against the 100 MHz wall clock).  Safe beside running proofs; bench.py samples it over its timed region."""
    out = (ctypes.c_uint64 * 2)()
    _check(lib().vgpu_shader_clock_probe(ctypes.c_int32(device), ctypes.c_uint32(iters), out))
    return float(out[0]) / float(out[1]) * 1e8


def poseidon_round_constants(seed=0x56414C494441, source="splitmix", **cli_switches):
    """480 Poseidon-16 round constants from SplitMix64(seed), rejection-sampled 31-bit values < p.

    The reference draws them from thread_rng (basic/tests/test_prover.rs:422); they are configuration
    input here — the same array must be given to the prover and the verifier.
    source="cli": the constants of the reference's CLI instead — Pcg64 seeded by Seeder::from("validia seed")
    (basic/src/bin/valida.rs:364-365), restated in valida_amd/cli_constants.py (switches: raw_monty, sip_adj0).
    """
    if source == "cli":
        from .cli_constants import cli_poseidon_round_constants

        return cli_poseidon_round_constants(**cli_switches)
    if source != "splitmix":
        raise ValueError("source must be 'splitmix' or 'cli'")
    out = []
    x = seed & 0xFFFFFFFFFFFFFFFF
    while len(out) < 480:
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        v = (z >> 33) & 0x7FFFFFFF
        if v < P:
            out.append(v)
    return np.array(out, dtype=np.uint32)


class OplogDesc(ctypes.Structure):  # vgpu_oplog_desc_t
    _fields_ = [
        ("struct_size", ctypes.c_uint64),  # = sizeof(vgpu_oplog_desc_t); vgpu_workload_oplog fills it in
        ("cpu", ctypes.c_void_p), ("n_cpu", ctypes.c_uint64),
        ("mem", ctypes.c_void_p), ("n_mem", ctypes.c_uint64),
        ("alu", ctypes.c_void_p * 4), ("n_alu", ctypes.c_uint64 * 4),
        ("static_cells", ctypes.c_void_p), ("n_static", ctypes.c_uint64),
        ("rom_len", ctypes.c_uint32),
        ("alu2", ctypes.c_void_p * 4), ("n_alu2", ctypes.c_uint64 * 4),  # mul, div, shift, com
        ("output", ctypes.c_void_p), ("n_output", ctypes.c_uint64),      # OutputChip::values as (clk, byte)
    ]


def decode_interaction_words(w):
    """Inverse of the neutral interaction word image (vgpu_machine_interaction_words)."""
    w = [int(x) for x in w]
    pos = 1

    def vcol():
        nonlocal pos
        nt, const = w[pos], w[pos + 1]
        pos += 2
        terms = []
        for _ in range(nt):
            terms.append((w[pos], w[pos + 1], w[pos + 2]))
            pos += 3
        return (const, terms)

    out = []
    for _ in range(w[0]):
        send, glob, bus, nf = w[pos:pos + 4]
        pos += 4
        count = vcol()
        out.append({"send": bool(send), "global": bool(glob), "bus": bus, "count": count, "fields": [vcol() for _ in range(nf)]})
    assert pos == len(w)
    return out


GENERATED_CHIPS = tuple(range(14))  # every BasicMachine chip has a log-driven device trace generator


class Workload:
    """Synthetic workload: fib_program(n) run on the BasicMachine, all chip traces generated (host)."""

    def __init__(self, handle):
        self._h = handle
        st = (ctypes.c_uint64 * 8)()
        lib().vgpu_workload_stats(self._h, st)
        self.cycles, self.cpu_ops, self.mem_ops, self.add_ops, self.result, self.program_len, self.cpu_height = [int(v) for v in st[:7]]

    @classmethod
    def fib(cls, n):
        h = ctypes.c_void_p()
        _check(lib().vgpu_workload_fib(ctypes.c_uint32(n), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def alu(cls, iters):
        """ALU-heavy loop (workload C4): 4 + 9 * iters + 1 cycles, exercising add / sub / bitwise / lt chips."""
        h = ctypes.c_void_p()
        _check(lib().vgpu_workload_alu(ctypes.c_uint32(iters), ctypes.byref(h)))
        return cls(h)

    @classmethod
    def named(cls, name):
        """One of the reference's other pinned prover programs: left_imm_ops, signed_inequality, loadfp, static_data; or "mixed_ops[:iters]", the
        synthetic program that keeps every chip busy (mul, div, shift, com, output included)."""
        h = ctypes.c_void_p()
        _check(lib().vgpu_workload_named(name.encode(), ctypes.byref(h)))
        return cls(h)

    def cell(self, addr):
        v = ctypes.c_uint32()
        _check(lib().vgpu_workload_cell(self._h, ctypes.c_uint32(addr), ctypes.byref(v)))
        return int(v.value)

    def main_trace(self, chip):
        data, h, w = c_u32p(), ctypes.c_uint64(), ctypes.c_uint64()
        _check(lib().vgpu_workload_main_trace(self._h, ctypes.c_uint32(chip), ctypes.byref(data), ctypes.byref(h), ctypes.byref(w)))
        return np.ctypeslib.as_array(data, shape=(h.value, w.value)).copy()  # the workload owns the buffer

    def main_traces(self):
        return [self.main_trace(i) for i in range(NUM_CHIPS)]

    def oplog(self):
        """The VM's operation logs (vgpu_oplog_desc_t; buffers owned by this workload)."""
        d = OplogDesc()
        lib().vgpu_workload_oplog(self._h, ctypes.byref(d))
        d._owner = self
        return d

    def preprocessed(self):
        """[(chip index, matrix)] in chip order: program ROM, range table."""
        out = []
        for k in range(2):
            chip, data, h, w = ctypes.c_uint32(), c_u32p(), ctypes.c_uint64(), ctypes.c_uint64()
            _check(lib().vgpu_workload_preprocessed(self._h, ctypes.c_uint32(k), ctypes.byref(chip), ctypes.byref(data), ctypes.byref(h), ctypes.byref(w)))
            out.append((chip.value, np.ctypeslib.as_array(data, shape=(h.value, w.value)).copy()))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_workload_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


class Machine:
    """Ordered chips with compiled constraint programs (basic/src/lib.rs:151-166)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def basic(cls):
        h = ctypes.c_void_p()
        _check(lib().vgpu_machine_basic(ctypes.byref(h)))
        return cls(h)

    @classmethod
    def basic_via_ffi(cls):
        """The same chips captured through the vgpu_air_* FFI (what a foreign host does): interpreted AIR programs."""
        h = ctypes.c_void_p()
        _check(lib().vgpu_machine_basic_via_ffi(ctypes.byref(h)))
        return cls(h)

    @property
    def num_chips(self):
        return int(lib().vgpu_machine_num_chips(self._h))

    def chip_info(self, chip):
        out = (ctypes.c_uint32 * 8)()
        _check(lib().vgpu_machine_chip_info(self._h, ctypes.c_uint32(chip), out))
        keys = ["width", "preprocessed_width", "interactions", "log_quotient_degree", "constraints", "instructions", "registers", "max_degree"]
        return dict(zip(keys, [int(v) for v in out]))

    def interactions(self, chip):
        """Chip::all_interactions of `chip` as a list of dicts (send, global, bus, count, fields); a virtual column is
        (constant, [(is_preprocessed, column, weight), ...])."""
        L = lib()
        L.vgpu_machine_interaction_words.restype = ctypes.c_int64
        n = L.vgpu_machine_interaction_words(self._h, ctypes.c_uint32(chip), None, ctypes.c_uint64(0))
        if n < 0:
            _check(int(n))
        w = np.zeros(int(n), dtype=np.uint32)
        L.vgpu_machine_interaction_words(self._h, ctypes.c_uint32(chip), w.ctypes.data_as(c_u32p), ctypes.c_uint64(w.size))
        return decode_interaction_words(w)

    def eval_constraints(self, chip, main_local, main_next, prep_local=None, prep_next=None, is_first=0, is_last=0, is_transition=1):
        info = self.chip_info(chip)
        pl = np.zeros(max(1, info["preprocessed_width"]), dtype=np.uint32) if prep_local is None else prep_local
        pn = np.zeros(max(1, info["preprocessed_width"]), dtype=np.uint32) if prep_next is None else prep_next
        a, ap = _u32(main_local)
        b, bp = _u32(main_next)
        c, cp = _u32(pl)
        d, dp = _u32(pn)
        out = np.zeros(max(1, info["constraints"]), dtype=np.uint32)
        n = lib().vgpu_machine_eval_constraints(self._h, ctypes.c_uint32(chip), ap, bp, cp, dp, ctypes.c_uint32(is_first), ctypes.c_uint32(is_last),
                                                ctypes.c_uint32(is_transition), out.ctypes.data_as(c_u32p), ctypes.c_uint32(out.size))
        if n < 0:
            _check(n)
        return out[:n]

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_machine_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


class Challenger:
    """DuplexChallenger<BabyBear, Poseidon16, 16> (host)."""

    def __init__(self, rc):
        self._rc, rcp = _u32(rc)
        self._h = ctypes.c_void_p()
        _check(lib().vgpu_challenger_new(rcp, ctypes.byref(self._h)))

    def observe(self, values):
        v, vp = _u32(values)
        lib().vgpu_challenger_observe(self._h, vp, ctypes.c_uint64(v.size))

    def sample(self, n=1):
        out = np.zeros(n, dtype=np.uint32)
        lib().vgpu_challenger_sample(self._h, out.ctypes.data_as(c_u32p), ctypes.c_uint64(n))
        return out

    def sample_bits(self, bits):
        return int(lib().vgpu_challenger_sample_bits(self._h, ctypes.c_uint32(bits)))

    def grind(self, bits):
        return int(lib().vgpu_challenger_grind(self._h, ctypes.c_uint32(bits)))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_challenger_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


def poseidon16_permute(rc, state):
    r, rp = _u32(rc)
    s = np.array(state, dtype=np.uint32)
    lib().vgpu_poseidon16_permute(rp, s.ctypes.data_as(c_u32p))
    return s


class _PinnedBlock:
    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        _check(lib().vgpu_host_alloc(ctypes.c_uint64(max(1, nbytes)), ctypes.byref(self.ptr)))

    def __del__(self):
        if getattr(self, "ptr", None):
            try:
                lib().vgpu_host_free(self.ptr)
            except TypeError:  # interpreter shutdown
                pass
            self.ptr = None


def pinned_empty(shape, dtype=np.uint32):
    """numpy array in page-locked host memory (vgpu_host_alloc): uploads from it are plain DMA.  The memory lives as long as the array
    (or any view of it) does."""
    shape = (shape,) if isinstance(shape, int) else tuple(shape)
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) if shape else 1
    blk = _PinnedBlock(n * dt.itemsize)
    buf = (ctypes.c_char * max(1, n * dt.itemsize)).from_address(blk.ptr.value)
    buf._vgpu_block = blk  # the ctypes buffer (numpy's .base) keeps the allocation alive
    return np.frombuffer(buf, dtype=dt, count=n).reshape(shape)


def pinned_copy(a):
    out = pinned_empty(a.shape, a.dtype)
    out[...] = a
    return out


class DeviceOplog:
    def __init__(self, prover, handle, nbytes):
        self._prover, self._h, self.nbytes = prover, handle, nbytes

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_oplog_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


class DeviceTrace:
    def __init__(self, prover, handle, shape):
        self._prover, self._h, self.shape = prover, handle, shape

    def download(self):
        """Canonical row-major copy (what the reference's generate_trace would have returned)."""
        out = np.zeros(self.shape, dtype=np.uint32)
        _check(lib().vgpu_trace_download(self._prover._h, self._h, out.ctypes.data_as(c_u32p), ctypes.c_uint64(out.size)))
        return out

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_trace_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


class ProverData:
    """Pcs::ProverData — committed LDEs + Merkle tree, resident in HBM."""

    def __init__(self, prover, handle, root, shapes):
        self._prover, self._h, self.root, self._shapes = prover, handle, root, shapes

    def lde(self, idx):
        """get_ldes()[idx] in committed (bit-reversed) row order, canonical values."""
        h, w = self._shapes[idx]
        H = h << self._prover.log_blowup
        out = np.zeros((H, w), dtype=np.uint32)
        _check(lib().vgpu_pdata_lde(self._prover._h, self._h, ctypes.c_uint32(idx), out.ctypes.data_as(c_u32p), ctypes.c_uint64(out.size)))
        return out

    def lde_view(self, idx):
        """pcs.get_ldes()[idx] without a copy: (device pointer, height, width, stride, log_blowup) of the committed LDE in HBM."""

        class _View(ctypes.Structure):  # vgpu_lde_view_t
            _fields_ = [("data", ctypes.c_void_p), ("height", ctypes.c_uint64), ("width", ctypes.c_uint64), ("stride", ctypes.c_uint64), ("log_blowup", ctypes.c_uint32)]

        v = _View()
        _check(lib().vgpu_pdata_lde_view(self._h, ctypes.c_uint32(idx), ctypes.byref(v)))
        return {"data": v.data, "height": int(v.height), "width": int(v.width), "stride": int(v.stride), "log_blowup": int(v.log_blowup)}

    @property
    def num_matrices(self):
        lib().vgpu_pdata_num_matrices.restype = ctypes.c_uint32
        return int(lib().vgpu_pdata_num_matrices(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_pdata_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


class Proof:
    def __init__(self, handle):
        self._h = handle
        n = int(lib().vgpu_proof_len(handle))
        self.words = np.ctypeslib.as_array(lib().vgpu_proof_words(handle), shape=(n,)).copy()
        ms = (ctypes.c_double * 11)()
        lib().vgpu_proof_phase_ms(handle, ms)
        keys = ["ingest", "commit_main", "perm", "commit_perm", "quotient", "commit_quotient", "open_values", "open_reduce", "fri", "queries", "total"]
        self.phase_ms = dict(zip(keys, [float(v) for v in ms]))
        t = (ctypes.c_uint32 * 33)()
        lib().vgpu_proof_transcript(handle, t)
        self.transcript = np.array(t, dtype=np.uint32)

    def bytes(self):
        return self.words.tobytes()

    def cbor(self, flags=0):
        """The proof as the reference writes it to disk: CBOR of the serde-derived MachineProof (see proof_cbor)."""
        return proof_cbor(self.words, flags)

    def _dbg(self, fn, chip):
        n = fn(self._h, ctypes.c_uint32(chip), None, ctypes.c_uint64(0))
        if n < 0:
            raise IndexError(chip)
        out = np.zeros(n, dtype=np.uint32)
        fn(self._h, ctypes.c_uint32(chip), out.ctypes.data_as(c_u32p), ctypes.c_uint64(n))
        return out

    def debug_perm_trace(self, chip):
        return self._dbg(lib().vgpu_proof_debug_perm_trace, chip)

    def debug_quotient(self, chip):
        return self._dbg(lib().vgpu_proof_debug_quotient, chip)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_proof_free(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None


CBOR_CANONICAL_FIELDS, CBOR_PLAIN_DIGESTS = 1, 2


def proof_cbor(words, flags=0):
    """CBOR bytes of a proof given as VPF1 words (ciborium image of MachineProof; host-only, no device needed)."""
    w = np.ascontiguousarray(words, dtype=np.uint32)
    L = lib()
    L.vgpu_proof_cbor.restype = ctypes.c_int64
    n = L.vgpu_proof_cbor(w.ctypes.data_as(c_u32p), ctypes.c_uint64(w.size), ctypes.c_uint32(flags), None, ctypes.c_uint64(0))
    if n < 0:
        _check(int(n))
    buf = (ctypes.c_uint8 * int(n))()
    L.vgpu_proof_cbor(w.ctypes.data_as(c_u32p), ctypes.c_uint64(w.size), ctypes.c_uint32(flags), buf, ctypes.c_uint64(int(n)))
    return bytes(buf)


def proof_from_cbor(data):
    """VPF1 proof words from the CBOR image of a MachineProof (vgpu_proof_from_cbor: what `ciborium::from_reader` does on the verifier's
    side; host-only).  Either setting of the two encoding switches is accepted.  Raises VgpuError on malformed input."""
    b = bytes(data)
    buf = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")
    L = lib()
    L.vgpu_proof_from_cbor.restype = ctypes.c_int64
    n = L.vgpu_proof_from_cbor(buf, ctypes.c_uint64(len(b)), None, ctypes.c_uint64(0))
    if n < 0:
        _check(int(n))
    out = np.zeros(int(n), dtype=np.uint32)
    L.vgpu_proof_from_cbor(buf, ctypes.c_uint64(len(b)), out.ctypes.data_as(c_u32p), ctypes.c_uint64(int(n)))
    return out


def proof_from_cbor_ex(data, bare_is_montgomery=False):
    """(proof words, forms seen) — vgpu_proof_from_cbor_ex: the decoder with the one reading the image itself cannot decide (a bare integer as
    the Montgomery word) selectable, and a report of the encodings met: bit 1 field as {"value": m}, 2 field bare, 4 digest as Hash struct, 8 plain."""
    b = bytes(data)
    buf = (ctypes.c_uint8 * max(1, len(b))).from_buffer_copy(b if b else b"\0")
    L = lib()
    L.vgpu_proof_from_cbor_ex.restype = ctypes.c_int64
    seen = ctypes.c_uint32(0)
    fl = ctypes.c_uint32(1 if bare_is_montgomery else 0)
    n = L.vgpu_proof_from_cbor_ex(buf, ctypes.c_uint64(len(b)), fl, None, ctypes.c_uint64(0), ctypes.byref(seen))
    if n < 0:
        _check(int(n))
    out = np.zeros(int(n), dtype=np.uint32)
    L.vgpu_proof_from_cbor_ex(buf, ctypes.c_uint64(len(b)), fl, out.ctypes.data_as(c_u32p), ctypes.c_uint64(int(n)), ctypes.byref(seen))
    return out, int(seen.value)


class Comm:
    """RCCL communicator owned by the library (one process per GPU): the path's one collective without Python in the loop."""

    def __init__(self, prover, unique_id, rank, world):
        self._prover, self.rank, self.world = prover, rank, world
        self._h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(lib().vgpu_comm_init(prover._h, buf, ctypes.c_uint32(rank), ctypes.c_uint32(world), ctypes.byref(self._h)))

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * 128)()
        _check(lib().vgpu_comm_unique_id(buf))
        return bytes(buf)

    def set_timeout_ms(self, timeout_ms):
        """Deadline of every collective of this communicator (vgpu_comm_set_timeout_ms): past it the communicator is aborted and the call fails."""
        _check(lib().vgpu_comm_set_timeout_ms(self._h, ctypes.c_uint32(int(timeout_ms))))

    def commit_batches_sharded(self, traces, coset_shifts=None):
        """This rank's share of a sharded pcs.commit_batches over RCCL; returns the (common) root."""
        arr = (ctypes.c_void_p * len(traces))(*[t._h for t in traces])
        root = np.zeros(8, dtype=np.uint32)
        sh = None
        if coset_shifts is not None:
            shv, sh = _u32(coset_shifts)
        _check(lib().vgpu_commit_batches_sharded(self._prover._h, self._h, arr, ctypes.c_uint32(len(traces)), sh, root.ctypes.data_as(c_u32p)))
        return root

    def prove_sharded(self, main, preprocessed, log_min_sharded=12):
        """This rank's share of ONE proof sharded over the communicator's ranks (vgpu_prove_sharded); every rank passes the same traces
        (uploaded through its own prover) and receives the proof vgpu_prove gives on one GPU."""
        arr = (ctypes.c_void_p * len(main))(*[t._h for t in main])
        chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
        parr = (ctypes.c_void_p * max(1, len(preprocessed)))(*[t._h for _, t in preprocessed])
        h = ctypes.c_void_p()
        _check(lib().vgpu_prove_sharded(self._prover._h, self._h, arr, ctypes.c_uint32(len(main)), chips, parr, ctypes.c_uint32(len(preprocessed)),
                                        ctypes.c_uint32(log_min_sharded), ctypes.byref(h)))
        return Proof(h)

    def allgather_roots(self, words):
        w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1)
        out = np.zeros(w.size * self.world, dtype=np.uint32)
        _check(lib().vgpu_comm_allgather_roots(self._h, w.ctypes.data_as(c_u32p), ctypes.c_uint32(w.size), out.ctypes.data_as(c_u32p)))
        return out.reshape(self.world, w.size)

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_comm_destroy(self._h)
            except TypeError:  # interpreter shutdown
                pass
            self._h = None


def verify_multi_batches(commits, heights, widths, points, values, proof_words, challenger, rc, log_blowup=1, num_queries=40, pow_bits=8,
                         hash_kind=HASH_KECCAK256, observe_final_poly=False):
    """pcs.verify_multi_batches on the host (vgpu_verify_multi_batches): commits[r] = 8 words; heights[r][i], widths[r][i] =
    Dimensions of matrix i of round r; points[r][i] = list of Ext5; values = flat words as open_multi_batches returned them (or the
    nested (width, 5) arrays).  Returns None if accepted, else the rejection message."""
    cfg = VgpuConfig()
    cfg.log_blowup, cfg.num_queries, cfg.pow_bits, cfg.hash_kind, cfg.observe_final_poly = log_blowup, num_queries, pow_bits, int(hash_kind), int(observe_final_poly)
    r = np.ascontiguousarray(rc, dtype=np.uint32)
    ctypes.memmove(cfg.poseidon_rc, r.ctypes.data, 480 * 4)
    c, cp = _u32(np.concatenate([np.asarray(x, dtype=np.uint32) for x in commits]))
    nm = np.array([len(h) for h in heights], dtype=np.uint32)
    hs = np.array([h for rnd in heights for h in rnd], dtype=np.uint64)
    ws = np.array([w for rnd in widths for w in rnd], dtype=np.uint32)
    npts = np.array([len(p) for rnd in points for p in rnd], dtype=np.uint32)
    pts = np.array([w for rnd in points for p in rnd for z in p for w in z], dtype=np.uint32)
    if isinstance(values, (list, tuple)):
        values = np.concatenate([v.ravel() for rnd in values for mat in rnd for v in mat])
    v, vp = _u32(values)
    pw, pwp = _u32(proof_words)
    code = lib().vgpu_verify_multi_batches(ctypes.byref(cfg), cp, ctypes.c_uint32(len(commits)), nm.ctypes.data_as(c_u32p), hs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                           ws.ctypes.data_as(c_u32p), npts.ctypes.data_as(c_u32p), pts.ctypes.data_as(c_u32p), vp, ctypes.c_uint64(v.size), pwp,
                                           ctypes.c_uint64(pw.size), challenger._h)
    return None if code == 0 else lib().vgpu_last_error().decode()


def _config(rc, log_blowup=1, num_queries=40, pow_bits=8, hash_kind=HASH_KECCAK256, observe_final_poly=False):
    cfg = VgpuConfig()
    cfg.log_blowup, cfg.num_queries, cfg.pow_bits, cfg.hash_kind, cfg.observe_final_poly = log_blowup, num_queries, pow_bits, int(hash_kind), int(observe_final_poly)
    r = np.ascontiguousarray(rc, dtype=np.uint32)
    ctypes.memmove(cfg.poseidon_rc, r.ctypes.data, 480 * 4)
    return cfg


def verify(machine, rc, proof_words, preprocessed_commit=None, **cfg_kw):
    """Machine::verify on the host (vgpu_verify; no device needed): None if the proof is accepted, else the rejection message.
    preprocessed_commit: 8 words (host_commit_root of the preprocessed traces, as the reference's verifier recomputes it), None for a machine without."""
    cfg = _config(rc, **cfg_kw)
    pw, pwp = _u32(proof_words)
    pc = None
    if preprocessed_commit is not None:
        pcv, pc = _u32(preprocessed_commit)
    code = lib().vgpu_verify(ctypes.byref(cfg), machine._h, pc, pwp, ctypes.c_uint64(pw.size))
    return None if code == 0 else lib().vgpu_last_error().decode()


def host_commit_root(matrices, rc, coset_shifts=None, **cfg_kw):
    """pcs.commit_batches on the host (vgpu_host_commit_root): the root vgpu_commit_batches gives on the device; for the small matrices a
    verifier commits itself."""
    cfg = _config(rc, **cfg_kw)
    keep = [np.ascontiguousarray(m, dtype=np.uint32) for m in matrices]
    ptrs = (c_u32p * len(keep))(*[k.ctypes.data_as(c_u32p) for k in keep])
    hs = (ctypes.c_uint64 * len(keep))(*[k.shape[0] for k in keep])
    ws = (ctypes.c_uint64 * len(keep))(*[k.shape[1] for k in keep])
    sh = None
    if coset_shifts is not None:
        shv, sh = _u32(coset_shifts)
    root = np.zeros(8, dtype=np.uint32)
    _check(lib().vgpu_host_commit_root(ctypes.byref(cfg), ptrs, hs, ws, ctypes.c_uint32(len(keep)), sh, root.ctypes.data_as(c_u32p)))
    return root


def commit_batches_sharded_local(provers, matrices, coset_shifts=None):
    """One commitment round sharded over len(provers) prover contexts of this process (vgpu_commit_batches_sharded_local):
    returns the root, which equals provers[0].commit_batches(..).root."""
    keep = [[p.upload(m) for m in matrices] for p in provers]
    arr = (ctypes.c_void_p * (len(provers) * len(matrices)))(*[t._h for row in keep for t in row])
    parr = (ctypes.c_void_p * len(provers))(*[p._h for p in provers])
    root = np.zeros(8, dtype=np.uint32)
    sh = None
    if coset_shifts is not None:
        shv, sh = _u32(coset_shifts)
    _check(lib().vgpu_commit_batches_sharded_local(parr, ctypes.c_uint32(len(provers)), arr, ctypes.c_uint32(len(matrices)), sh, root.ctypes.data_as(c_u32p)))
    return root


def upload_replicated(provers, main_traces, preprocessed):
    """The traces of one proof uploaded through every prover context (a sharded proof replicates the traces): (main, prep) handles for
    prove_sharded_local(.., uploaded=..)."""
    return [[p.upload(m) for m in main_traces] for p in provers], [[p.upload(m) for _, m in preprocessed] for p in provers]


def sharded_trace_is_split(world, height, log_blowup=1, log_min_sharded=12):
    """vgpu_sharded_trace_is_split: whether a chip of this trace height hands in row ranges (its LDE is sharded) in the row-range form of a sharded proof."""
    f = lib().vgpu_sharded_trace_is_split
    f.restype = ctypes.c_uint32
    return bool(f(ctypes.c_uint32(world), ctypes.c_uint32(log_blowup), ctypes.c_uint32(log_min_sharded), ctypes.c_uint64(int(height))))


def row_ranges(main_traces, rank, world, log_blowup=1, log_min_sharded=12):
    """Rank `rank`'s inputs of a sharded proof in ROW-RANGE form: (matrices, full_heights) — the rows [rank n / world, (rank + 1) n / world) of every chip
    whose LDE is sharded, the whole trace of every other chip."""
    out, full = [], []
    for m in main_traces:
        n = m.shape[0]
        full.append(n)
        if sharded_trace_is_split(world, n, log_blowup, log_min_sharded):
            out.append(np.ascontiguousarray(m[rank * (n // world):(rank + 1) * (n // world)]))
        else:
            out.append(m)
    return out, full


def upload_row_ranges(provers, main_traces, preprocessed, log_min_sharded=12):
    """The ROW-RANGE inputs of one sharded proof uploaded through the prover contexts standing in for the ranks: context r gets only its rows of
    every sharded chip.  For prove_sharded_rows_local(.., uploaded=..)."""
    keep, full = [], None
    for r, p in enumerate(provers):
        mats, full = row_ranges(main_traces, r, len(provers), p.log_blowup, log_min_sharded)
        keep.append([p.upload(m) for m in mats])
    return keep, [[p.upload(m) for _, m in preprocessed] for p in provers], full


def prove_sharded_rows_local(provers, main_traces, preprocessed, log_min_sharded=12, uploaded=None):
    """ONE proof over len(provers) prover contexts of this process with the TRACES sharded too (vgpu_prove_sharded_rows_local): context r uploads only
    its row range of every sharded chip (row_ranges).  Returns the Proof, word for word provers[0].prove(..) of the whole traces."""
    W = len(provers)
    keep, keep_p, full = uploaded if uploaded is not None else upload_row_ranges(provers, main_traces, preprocessed, log_min_sharded)
    arr = (ctypes.c_void_p * (W * len(main_traces)))(*[t._h for row in keep for t in row])
    fh = (ctypes.c_uint64 * len(main_traces))(*full)
    chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
    parr = (ctypes.c_void_p * max(1, W * len(preprocessed)))(*[t._h for row in keep_p for t in row])
    pv = (ctypes.c_void_p * W)(*[p._h for p in provers])
    h = ctypes.c_void_p()
    _check(lib().vgpu_prove_sharded_rows_local(pv, ctypes.c_uint32(W), arr, ctypes.c_uint32(len(main_traces)), fh, chips, parr, ctypes.c_uint32(len(preprocessed)),
                                               ctypes.c_uint32(log_min_sharded), ctypes.byref(h)))
    return Proof(h)


class VgpuFabric(ctypes.Structure):
    """vgpu_fabric_t (include/vgpu.h): the host's own transport for a sharded proof, two collective callbacks over host buffers."""
    ALL_GATHER = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, c_u32p, ctypes.c_uint64, c_u32p)
    ALL_TO_ALL = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(c_u32p), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(c_u32p),
                                  ctypes.POINTER(ctypes.c_uint64))
    _fields_ = [("struct_size", ctypes.c_uint32), ("timeout_ms", ctypes.c_uint32), ("user", ctypes.c_void_p), ("rank", ctypes.c_uint32), ("world", ctypes.c_uint32),
                ("all_gather", ALL_GATHER), ("all_to_all", ALL_TO_ALL)]


_ABANDONED_FABRICS = []  # fabrics whose callbacks the library gave up on at its deadline: a helper thread of the library may still be INSIDE one of their ctypes thunks, so they live as long as the process


class Fabric:
    """A caller-supplied fabric from two Python callables (what a Rust host would write against its own transport):
        all_gather(mine: np.uint32[n]) -> np.uint32[world, n]
        all_to_all(send: list of np.uint32 arrays or None per rank, recv_words: list of int) -> list of np.uint32 arrays (None where 0 words)
    Exceptions inside a callback become a non-zero status: the proof is abandoned on every rank (the library's failure protocol).
    timeout_ms > 0: the library bounds every callback (vgpu_fabric_t::timeout_ms) — it then calls them from a helper thread of its own."""

    def __init__(self, rank, world, all_gather, all_to_all, timeout_ms=0):
        self.rank, self.world = rank, world
        self.errors = []
        self._ag_py, self._a2a_py = all_gather, all_to_all  # looked up at call time: a test may wrap them

        def ag(_user, words, n, out):
            try:
                mine = np.ctypeslib.as_array(words, shape=(int(n),)).copy()
                got = np.ascontiguousarray(self._ag_py(mine), dtype=np.uint32).reshape(world, int(n))
                ctypes.memmove(out, got.ctypes.data, got.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001 - must not unwind into C
                self.errors.append(e)
                return 1

        def a2a(_user, send, send_words, recv, recv_words):
            try:
                sb = [np.ctypeslib.as_array(send[s], shape=(int(send_words[s]),)).copy() if send_words[s] else None for s in range(world)]
                rw = [int(recv_words[s]) for s in range(world)]
                got = self._a2a_py(sb, rw)
                for s in range(world):
                    if rw[s]:
                        g = np.ascontiguousarray(got[s], dtype=np.uint32).reshape(-1)
                        assert g.size == rw[s], "all_to_all delivered %d words from rank %d, expected %d" % (g.size, s, rw[s])
                        ctypes.memmove(recv[s], g.ctypes.data, g.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001
                self.errors.append(e)
                return 1

        self._ag, self._a2a = VgpuFabric.ALL_GATHER(ag), VgpuFabric.ALL_TO_ALL(a2a)  # kept alive with the object
        self.c = VgpuFabric(ctypes.sizeof(VgpuFabric), int(timeout_ms), None, rank, world, self._ag, self._a2a)

    @staticmethod
    def over_torch_distributed(dist, group=None, timeout_ms=0):
        """The exchanges through a torch.distributed process group on CPU tensors (gloo): all_gather, and pairwise isend / irecv."""
        import torch

        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def all_gather(mine):
            outs = [torch.zeros(mine.size, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(outs, torch.from_numpy(mine.view(np.int32).copy()), group=group)
            return np.stack([o.numpy().view(np.uint32) for o in outs])

        def all_to_all(send, recv_words):
            reqs, bufs = [], [None] * world
            for s in range(world):
                if s != rank and recv_words[s]:
                    bufs[s] = torch.zeros(recv_words[s], dtype=torch.int32)
                    reqs.append(dist.irecv(bufs[s], src=s, group=group))
            keep = []
            for s in range(world):
                if s != rank and send[s] is not None:
                    t = torch.from_numpy(send[s].view(np.int32))
                    keep.append(t)
                    reqs.append(dist.isend(t, dst=s, group=group))
            for r in reqs:
                r.wait()
            return [None if b is None else b.numpy().view(np.uint32) for b in bufs]

        return Fabric(rank, world, all_gather, all_to_all, timeout_ms=timeout_ms)

    def selftest(self, n_words=17, fail_rank=0xFFFFFFFF):
        """vgpu_fabric_selftest: None if this rank's exchanges were all correct, else the library's message (a failing peer included)."""
        code = lib().vgpu_fabric_selftest(ctypes.byref(self.c), ctypes.c_uint32(n_words), ctypes.c_uint32(fail_rank))
        return None if code == 0 else lib().vgpu_last_error().decode()

    def prove_sharded(self, prover, main, preprocessed, log_min_sharded=12, full_heights=None):
        """This rank's share of ONE proof sharded over the fabric's ranks (vgpu_prove_sharded_fabric): every rank passes the same traces,
        uploaded through its own prover, and receives the proof vgpu_prove gives on one GPU.  full_heights given: the ROW-RANGE form
        (vgpu_prove_sharded_rows_fabric) — `main` holds this rank's rows of every sharded chip (row_ranges)."""
        arr = (ctypes.c_void_p * len(main))(*[t._h for t in main])
        chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
        parr = (ctypes.c_void_p * max(1, len(preprocessed)))(*[t._h for _, t in preprocessed])
        h = ctypes.c_void_p()
        try:
            if full_heights is not None:
                fh = (ctypes.c_uint64 * len(main))(*[int(x) for x in full_heights])
                _check(lib().vgpu_prove_sharded_rows_fabric(prover._h, ctypes.byref(self.c), arr, ctypes.c_uint32(len(main)), fh, chips, parr, ctypes.c_uint32(len(preprocessed)),
                                                            ctypes.c_uint32(log_min_sharded), ctypes.byref(h)))
                return Proof(h)
            _check(lib().vgpu_prove_sharded_fabric(prover._h, ctypes.byref(self.c), arr, ctypes.c_uint32(len(main)), chips, parr, ctypes.c_uint32(len(preprocessed)),
                                                   ctypes.c_uint32(log_min_sharded), ctypes.byref(h)))
            return Proof(h)
        except VgpuError as e:
            # only a callback ABANDONED at the deadline (VGPU_ERR_FABRIC with the library's "callback is abandoned", csrc/host/fabric.hpp) leaves a library
            # thread inside one of this object's thunks: those must stay valid for the life of the process.  Argument errors, dead peers and failpoints
            # leave nothing behind, and a host that retries must not pin every Fabric it ever made.
            if self.c.timeout_ms and e.code == -6 and "callback is abandoned" in str(e) and not any(f is self for f in _ABANDONED_FABRICS):
                _ABANDONED_FABRICS.append(self)
            raise


def prove_sharded_local(provers, main_traces, preprocessed, log_min_sharded=12, uploaded=None):
    """ONE proof over len(provers) prover contexts of this process standing in for the ranks (vgpu_prove_sharded_local): main_traces are
    host matrices (chip order), preprocessed a list of (chip, host matrix); every context gets its own upload (the traces are replicated,
    the LDEs / trees / FRI layers are sharded) unless `uploaded` = upload_replicated(..) hands them in.  Returns the Proof, word for word
    provers[0].prove(..) of the same traces."""
    keep_m, keep_p = uploaded if uploaded is not None else upload_replicated(provers, main_traces, preprocessed)
    arr = (ctypes.c_void_p * (len(provers) * len(main_traces)))(*[t._h for row in keep_m for t in row])
    chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
    parr = (ctypes.c_void_p * max(1, len(provers) * len(preprocessed)))(*[t._h for row in keep_p for t in row])
    pv = (ctypes.c_void_p * len(provers))(*[p._h for p in provers])
    h = ctypes.c_void_p()
    _check(lib().vgpu_prove_sharded_local(pv, ctypes.c_uint32(len(provers)), arr, ctypes.c_uint32(len(main_traces)), chips, parr, ctypes.c_uint32(len(preprocessed)),
                                          ctypes.c_uint32(log_min_sharded), ctypes.byref(h)))
    return Proof(h)


class Ticket:
    """An outstanding asynchronous proof (vgpu_prove_async); keeps its inputs alive until waited for."""

    def __init__(self, handle, keep):
        self._h, self._keep = handle, keep

    def wait(self):
        h = ctypes.c_void_p()
        t, self._h = self._h, None
        _check(lib().vgpu_ticket_wait(t, ctypes.byref(h)))
        self._keep = None
        return Proof(h)

    def __del__(self):
        if getattr(self, "_h", None):  # never leave a worker thread using freed traces behind
            try:
                self.wait()
            except Exception:
                pass


class Prover:
    """One MI355X: StarkConfig + Machine -> prove().  Mirrors machine.prove(&config)."""

    def __init__(self, machine, rc, device=0, log_blowup=1, num_queries=40, pow_bits=8, observe_final_poly=False, interpret_air=False, hash_kind=HASH_KECCAK256):
        cfg = VgpuConfig()
        cfg.device, cfg.log_blowup, cfg.num_queries, cfg.pow_bits = device, log_blowup, num_queries, pow_bits
        cfg.hash_kind, cfg.observe_final_poly, cfg.interpret_air = int(hash_kind), int(observe_final_poly), int(interpret_air)
        rc = np.ascontiguousarray(rc, dtype=np.uint32)
        assert rc.size == 480
        ctypes.memmove(cfg.poseidon_rc, rc.ctypes.data, 480 * 4)
        self.machine, self.rc, self.log_blowup, self.num_queries, self.pow_bits = machine, rc, log_blowup, num_queries, pow_bits
        self._h = ctypes.c_void_p()
        _check(lib().vgpu_prover_create(ctypes.byref(cfg), machine._h, ctypes.byref(self._h)))

    def upload(self, matrix):
        m = np.ascontiguousarray(matrix, dtype=np.uint32)
        assert m.ndim == 2
        h = ctypes.c_void_p()
        _check(lib().vgpu_trace_upload(self._h, m.ctypes.data_as(c_u32p), ctypes.c_uint64(m.shape[0]), ctypes.c_uint64(m.shape[1]), ctypes.byref(h)))
        return DeviceTrace(self, h, m.shape)

    def upload_oplog(self, desc):
        h = ctypes.c_void_p()
        _check(lib().vgpu_oplog_upload(self._h, ctypes.byref(desc), ctypes.byref(h)))
        nbytes = 48 * desc.n_cpu + 16 * desc.n_mem + 16 * sum(desc.n_alu) + 8 * desc.n_static
        log = DeviceOplog(self, h, int(nbytes))
        log._desc = desc
        return log

    def generate_trace(self, log, chip):
        """Chip::generate_trace of `chip` on the device, from the uploaded operation logs."""
        h = ctypes.c_void_p()
        _check(lib().vgpu_generate_trace(self._h, log._h, ctypes.c_uint32(chip), ctypes.byref(h)))
        hh, ww = ctypes.c_uint64(), ctypes.c_uint64()
        lib().vgpu_trace_shape(h, ctypes.byref(hh), ctypes.byref(ww))
        return DeviceTrace(self, h, (int(hh.value), int(ww.value)))

    def commit_batches(self, traces, coset_shifts=None):
        arr = (ctypes.c_void_p * len(traces))(*[t._h for t in traces])
        root = np.zeros(8, dtype=np.uint32)
        sh = None
        if coset_shifts is not None:
            shv, sh = _u32(coset_shifts)
        h = ctypes.c_void_p()
        _check(lib().vgpu_commit_batches(self._h, arr, ctypes.c_uint32(len(traces)), sh, root.ctypes.data_as(c_u32p), ctypes.byref(h)))
        return ProverData(self, h, root, [t.shape for t in traces])

    def generate_permutation_trace(self, chip, main, challenges, preprocessed=None):
        info = self.machine.chip_info(chip)
        n = main.shape[0]
        out = np.zeros((n, 5 * (info["interactions"] + 1)), dtype=np.uint32)
        ch, chp = _u32(challenges)
        cs = np.zeros(5, dtype=np.uint32)
        _check(lib().vgpu_perm_trace(self._h, ctypes.c_uint32(chip), main._h, preprocessed._h if preprocessed is not None else None, chp,
                                     out.ctypes.data_as(c_u32p), ctypes.c_uint64(out.size), cs.ctypes.data_as(c_u32p)))
        return out, cs

    def permutation_trace_device(self, chip, main, challenges, preprocessed=None):
        """generate_permutation_trace left in HBM: (DeviceTrace, cumulative_sum[5])."""
        ch, chp = _u32(challenges)
        cs = np.zeros(5, dtype=np.uint32)
        h = ctypes.c_void_p()
        _check(lib().vgpu_perm_trace_device(self._h, ctypes.c_uint32(chip), main._h, preprocessed._h if preprocessed is not None else None, chp, ctypes.byref(h),
                                            cs.ctypes.data_as(c_u32p)))
        hh, ww = ctypes.c_uint64(), ctypes.c_uint64()
        lib().vgpu_trace_shape(h, ctypes.byref(hh), ctypes.byref(ww))
        return DeviceTrace(self, h, (int(hh.value), int(ww.value))), cs

    def quotient(self, chip, main_pd, main_idx, perm_pd, perm_idx, perm_challenges, alpha, cumulative_sum, prep_pd=None, prep_idx=0):
        """quotient + decompose_and_flatten of one chip from its committed LDEs -> DeviceTrace of the chunk matrix (n x 10)."""
        a, ap = _u32(perm_challenges)
        b, bp = _u32(alpha)
        c, cp = _u32(cumulative_sum)
        h = ctypes.c_void_p()
        _check(lib().vgpu_quotient(self._h, ctypes.c_uint32(chip), prep_pd._h if prep_pd is not None else None, ctypes.c_uint32(prep_idx), main_pd._h,
                                   ctypes.c_uint32(main_idx), perm_pd._h, ctypes.c_uint32(perm_idx), ap, bp, cp, ctypes.byref(h)))
        hh, ww = ctypes.c_uint64(), ctypes.c_uint64()
        lib().vgpu_trace_shape(h, ctypes.byref(hh), ctypes.byref(ww))
        return DeviceTrace(self, h, (int(hh.value), int(ww.value)))

    def open_multi_batches(self, rounds, points, challenger):
        """pcs.open_multi_batches: rounds = [ProverData]; points[r][i] = list of Ext5 (5 words) for matrix i of round r.
        Returns (openings[r][i][p] as (width, 5) arrays, TwoAdicFriPcsProof words); advances `challenger`."""
        arr = (ctypes.c_void_p * len(rounds))(*[r._h for r in rounds])
        n_points = np.array([len(pts) for rnd in points for pts in rnd], dtype=np.uint32)
        flat = np.array([w for rnd in points for pts in rnd for z in pts for w in z], dtype=np.uint32)
        h = ctypes.c_void_p()
        _check(lib().vgpu_open_multi_batches(self._h, arr, ctypes.c_uint32(len(rounds)), n_points.ctypes.data_as(c_u32p), flat.ctypes.data_as(c_u32p), challenger._h,
                                             ctypes.byref(h)))
        L = lib()
        for f in (L.vgpu_opening_values_len, L.vgpu_opening_proof_len):
            f.restype = ctypes.c_uint64
        for f in (L.vgpu_opening_values, L.vgpu_opening_proof):
            f.restype = c_u32p
        try:
            nv, npf = int(L.vgpu_opening_values_len(h)), int(L.vgpu_opening_proof_len(h))
            vals = np.ctypeslib.as_array(L.vgpu_opening_values(h), shape=(nv,)).copy() if nv else np.zeros(0, np.uint32)
            proof = np.ctypeslib.as_array(L.vgpu_opening_proof(h), shape=(npf,)).copy()
        finally:
            L.vgpu_opening_free(h)
        out, pos = [], 0
        for r, rnd in enumerate(points):
            out.append([])
            for i, pts in enumerate(rnd):
                w = rounds[r]._shapes[i][1]
                out[r].append([])
                for _ in pts:
                    out[r][i].append(vals[pos:pos + 5 * w].reshape(w, 5))
                    pos += 5 * w
        assert pos == vals.size
        return out, proof

    def fri_fold(self, f, beta):
        f = np.ascontiguousarray(f, dtype=np.uint32)
        n = f.shape[0]
        b, bp = _u32(beta)
        out = np.zeros((n // 2, 5), dtype=np.uint32)
        _check(lib().vgpu_fri_fold(self._h, f.ctypes.data_as(c_u32p), ctypes.c_uint64(n), bp, out.ctypes.data_as(c_u32p)))
        return out

    def prove(self, main, preprocessed, debug=False, check=False):
        """main: list of DeviceTrace (chip order); preprocessed: list of (chip index, DeviceTrace).
        check: evaluate every constraint on the traces first (the reference's debug-build check_constraints)."""
        arr = (ctypes.c_void_p * len(main))(*[t._h for t in main])
        chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
        parr = (ctypes.c_void_p * max(1, len(preprocessed)))(*[t._h for _, t in preprocessed])
        h = ctypes.c_void_p()
        _check(lib().vgpu_prove(self._h, arr, ctypes.c_uint32(len(main)), chips, parr, ctypes.c_uint32(len(preprocessed)),
                                ctypes.c_uint32((1 if debug else 0) | (2 if check else 0)), ctypes.byref(h)))
        return Proof(h)

    def prove_async(self, main, preprocessed, keep=None):
        """Start Machine::prove on a host thread of the library; returns a Ticket (wait() -> Proof).  `keep`: further objects
        (e.g. the operation log the traces were generated from) kept alive until the ticket has been waited for."""
        arr = (ctypes.c_void_p * len(main))(*[t._h for t in main])
        chips = (ctypes.c_uint32 * max(1, len(preprocessed)))(*[c for c, _ in preprocessed])
        parr = (ctypes.c_void_p * max(1, len(preprocessed)))(*[t._h for _, t in preprocessed])
        h = ctypes.c_void_p()
        _check(lib().vgpu_prove_async(self._h, arr, ctypes.c_uint32(len(main)), chips, parr, ctypes.c_uint32(len(preprocessed)), ctypes.byref(h)))
        return Ticket(h, (main, preprocessed, self, keep))

    def set_prep_cache(self, on):
        """Keep the preprocessed commitment across proofs that hand in the same preprocessed traces (off by default: vgpu_prover_set_prep_cache)."""
        lib().vgpu_prover_set_prep_cache(self._h, ctypes.c_uint32(1 if on else 0))

    def set_profiling(self, on, only=None):
        """Per-kernel HIP-event timing on / off (resets the accumulators); only = time launches of this kernel name alone."""
        lib().vgpu_prover_set_profiling_filter(self._h, only.encode() if only else None)
        lib().vgpu_prover_set_profiling(self._h, ctypes.c_uint32(1 if on else 0))

    def profile(self):
        """{kernel: (launches, total_ms, algorithmic_bytes, algorithmic_valu_ops)} accumulated since set_profiling(True)."""
        n = lib().vgpu_prover_profile(self._h, None, ctypes.c_uint64(0))
        buf = ctypes.create_string_buffer(int(n) + 16)
        lib().vgpu_prover_profile(self._h, buf, ctypes.c_uint64(int(n) + 16))
        out = {}
        for line in buf.value.decode().splitlines():
            name, launches, ms, nbytes, ops = line.split()
            out[name] = (int(launches), float(ms), float(nbytes), float(ops))
        return out

    def trim(self):
        """Give the pool's cached free blocks back to the driver; returns the bytes freed."""
        lib().vgpu_prover_trim.restype = ctypes.c_uint64
        return int(lib().vgpu_prover_trim(self._h))

    def reset_memory_peak(self):
        """The pool's peak restarts from what is held now (vgpu_prover_memory_reset_peak)."""
        lib().vgpu_prover_memory_reset_peak(self._h)

    def memory(self):
        live, peak = ctypes.c_uint64(), ctypes.c_uint64()
        lib().vgpu_prover_memory(self._h, ctypes.byref(live), ctypes.byref(peak))
        return live.value, peak.value

    def __del__(self):
        if getattr(self, "_h", None):
            try:
                lib().vgpu_prover_destroy(self._h)
            except TypeError:  # interpreter shutdown: module globals already cleared
                pass
            self._h = None
