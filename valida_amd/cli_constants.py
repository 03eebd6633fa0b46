"""The ONE deterministic configuration the reference has: its CLI draws the Poseidon-16 round constants from

    let mut rng: Pcg64 = Seeder::from("validia seed").make_rng();          // basic/src/bin/valida.rs:364
    let perm16 = Perm16::new_from_rng(4, 22, mds16, &mut rng);             // :365

(the tests use thread_rng, basic/tests/test_prover.rs:422).  With these constants a proof file written by the real `valida prove` can be
checked by this backend's verifier (python -m valida_amd.verify_cli --constants cli).  None of the three crates is in /root/reference
(rand_pcg 0.3.1, rand_seeder 0.2.3, Plonky3's p3-poseidon are registry / git dependencies: Cargo.lock:1012-1052, Cargo.toml:24-41), so
every step below is a restatement of a published algorithm and is tagged with what pins it:

  [KAT]      reproduces a published known-answer vector, checked in tests/test_cli_constants_cpu.py
  [RECALL]   restated from memory of the crate's source; a keyword argument switches the alternative reading where one is plausible

  rand_pcg::Pcg64 = Lcg128Xsl64                                   [KAT]  the PCG reference suite's pcg64 vector (seed 42, stream 54)
      from_seed: state = le128(seed[0..16]), increment = le128(seed[16..32]) | 1, then state += increment; step            [RECALL]
  SipHash-2-4 (rand_seeder::SipHasher::new(): keys 0, 0)          [KAT]  the SipHash paper's vector (key 00..0f, input 00..0e)
  impl Hash for str: the bytes, then the byte 0xff                [RECALL] (core::hash)
  rand_seeder::SipHasher::into_rng / SipRng::next_u64             [RECALL] finish without the 0xff / d-rounds, then per output:
      v2 ^= adj; adj -= 0x11; 2 rounds; v0 ^ v1 ^ v2 ^ v3, adj starting at 0x13   (switch `sip_adj0`)
  Seeder::make_rng: 32 seed bytes = four next_u64, little-endian  [RECALL]
  Standard for BabyBear: next_u32() >> 1, rejected unless < p     [RECALL]; next_u32 of Pcg64 = low half of next_u64
  Poseidon::new_from_rng(4, 22, ..): 16 * 30 constants drawn in one sequence, round-major        [RECALL] (App. B7)
  the drawn u32 is stored RAW as the Montgomery representation    [RECALL] switch `raw_monty` (App. B7 RC_RAW_MONTY): the canonical
      constant is then value * 2^-32 mod p
"""
import numpy as np

P = 2013265921
M64 = (1 << 64) - 1
M128 = (1 << 128) - 1
PCG_MULTIPLIER = 0x2360ED051FC65DA44385DF649FCCF645  # PCG_DEFAULT_MULTIPLIER_128


class Pcg64:
    """rand_pcg::Lcg128Xsl64 (PCG XSL RR 128/64, LCG variant)."""

    def __init__(self, state, increment):
        self.increment = increment & M128
        self.state = (state + self.increment) & M128  # from_state_incr: "move away from the initial value"
        self._step()

    @classmethod
    def new(cls, state, stream):
        return cls(state, ((stream << 1) | 1) & M128)

    @classmethod
    def from_seed(cls, seed32):
        assert len(seed32) == 32
        w = [int.from_bytes(seed32[8 * i:8 * i + 8], "little") for i in range(4)]
        return cls(w[0] | (w[1] << 64), (w[2] | (w[3] << 64)) | 1)

    def _step(self):
        self.state = (self.state * PCG_MULTIPLIER + self.increment) & M128

    def next_u64(self):
        self._step()
        rot = self.state >> 122
        xsl = ((self.state >> 64) ^ self.state) & M64
        return ((xsl >> rot) | (xsl << (64 - rot))) & M64 if rot else xsl

    def next_u32(self):
        return self.next_u64() & 0xFFFFFFFF


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & M64


def _sip_round(v):
    v0, v1, v2, v3 = v
    v0 = (v0 + v1) & M64; v1 = _rotl(v1, 13); v1 ^= v0; v0 = _rotl(v0, 32)
    v2 = (v2 + v3) & M64; v3 = _rotl(v3, 16); v3 ^= v2
    v0 = (v0 + v3) & M64; v3 = _rotl(v3, 21); v3 ^= v0
    v2 = (v2 + v1) & M64; v1 = _rotl(v1, 17); v1 ^= v2; v2 = _rotl(v2, 32)
    return [v0, v1, v2, v3]


def _sip_absorb(data, k0, k1):
    v = [k0 ^ 0x736F6D6570736575, k1 ^ 0x646F72616E646F6D, k0 ^ 0x6C7967656E657261, k1 ^ 0x7465646279746573]
    n = len(data)
    for i in range(0, n - n % 8, 8):
        m = int.from_bytes(data[i:i + 8], "little")
        v[3] ^= m
        v = _sip_round(_sip_round(v))
        v[0] ^= m
    tail = int.from_bytes(data[n - n % 8:], "little")
    return v, ((n & 0xFF) << 56) | tail


def siphash24(data, k0=0, k1=0):
    v, b = _sip_absorb(bytes(data), k0, k1)
    v[3] ^= b
    v = _sip_round(_sip_round(v))
    v[0] ^= b
    v[2] ^= 0xFF
    for _ in range(4):
        v = _sip_round(v)
    return v[0] ^ v[1] ^ v[2] ^ v[3]


class SipRng:
    """rand_seeder::SipRng as SipHasher::into_rng leaves it."""

    def __init__(self, hashed_bytes, adj0=0x13):
        v, b = _sip_absorb(bytes(hashed_bytes), 0, 0)
        v[3] ^= b
        v = _sip_round(_sip_round(v))
        v[0] ^= b
        v = _sip_round(_sip_round(v))  # "d - c rounds"
        self.v, self.adj = v, adj0

    def next_u64(self):
        self.v[2] ^= self.adj
        self.adj = (self.adj - 0x11) & M64
        self.v = _sip_round(_sip_round(self.v))
        return self.v[0] ^ self.v[1] ^ self.v[2] ^ self.v[3]


def seeder_make_pcg64(text="validia seed", sip_adj0=0x13):
    """Seeder::from(text).make_rng::<Pcg64>()"""
    rng = SipRng(text.encode() + b"\xff", sip_adj0)
    seed = b"".join(rng.next_u64().to_bytes(8, "little") for _ in range(4))
    return Pcg64.from_seed(seed)


def cli_poseidon_round_constants(raw_monty=True, sip_adj0=0x13, text="validia seed"):
    """The 480 round constants of the reference CLI's Perm16 as CANONICAL field elements (the library's configuration input)."""
    rng = seeder_make_pcg64(text, sip_adj0)
    out = []
    while len(out) < 16 * 30:
        v = rng.next_u32() >> 1
        if v < P:
            out.append(v)
    if raw_monty:  # the drawn word IS the Montgomery representation: canonical = v * 2^-32 mod p
        rinv = pow(1 << 32, P - 2, P)
        out = [v * rinv % P for v in out]
    return np.array(out, dtype=np.uint32)
