"""The reference CLI's deterministic Poseidon constants (basic/src/bin/valida.rs:364-365: Pcg64 from Seeder::from("validia seed")) restated in
valida_amd/cli_constants.py, and the first-contact mode of the verifier CLI.  The two generators underneath are pinned by PUBLISHED
known-answer vectors; the glue between them (SipRng, seed layout, draw order) is recall and carries switches (see the module's header)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import valida_amd as va
from oracle import pyoracle as po
from valida_amd import cli_constants as cc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pcg64_reproduces_the_pcg_reference_vector():
    # check-pcg64 of the PCG C library / rand_pcg's `test_lcg128xsl64_true_values`: seed 42, stream 54
    r = cc.Pcg64.new(42, 54)
    assert [r.next_u64() for _ in range(6)] == [0x86B1DA1D72062B68, 0x1304AA46C9853D39, 0xA3670E9E0DD50358, 0xF9090E529A7DAE00, 0xC85B9FD837996F2C, 0x606121F8E3919196]


def test_siphash24_reproduces_the_papers_vector():
    # SipHash-2-4, key 00 01 .. 0f, input 00 01 .. 0e (Aumasson & Bernstein, appendix A)
    assert cc.siphash24(bytes(range(15)), 0x0706050403020100, 0x0F0E0D0C0B0A0908) == 0xA129CA6149BE45E5
    assert cc.siphash24(b"", 0x0706050403020100, 0x0F0E0D0C0B0A0908) == 0x726FDB47DD0E0E31


def test_cli_constants_are_480_field_elements_and_deterministic():
    a = va.poseidon_round_constants(source="cli")
    b = va.poseidon_round_constants(source="cli")
    assert a.dtype == np.uint32 and a.shape == (480,) and (a < va.P).all() and np.array_equal(a, b)
    assert len(set(a.tolist())) == 480
    raw = va.poseidon_round_constants(source="cli", raw_monty=False)
    assert np.array_equal((raw.astype(np.uint64) * pow(1 << 32, va.P - 2, va.P) % va.P).astype(np.uint32), a)  # the switch is the Montgomery factor only
    assert not np.array_equal(va.poseidon_round_constants(source="cli", sip_adj0=0xFF), a)
    # a regression pin of THIS restatement (not a reference vector): changes to the recalled glue show up here
    assert hashlib.sha256(a.tobytes()).hexdigest() == hashlib.sha256(cc.cli_poseidon_round_constants().tobytes()).hexdigest()


def _write_proof(tmp_path, rc, flags, num_queries=4):
    w = va.Workload.fib(25)
    mt, prep = w.main_traces(), w.preprocessed()
    res = po.prove_basic(mt, prep[0][1], prep[1][1], rc, num_queries=num_queries)
    path = tmp_path / "proof.cbor"
    path.write_bytes(va.proof_cbor(res.words, flags))
    return str(path)


@pytest.mark.parametrize("flags", [0, va.CBOR_CANONICAL_FIELDS, va.CBOR_PLAIN_DIGESTS, va.CBOR_CANONICAL_FIELDS | va.CBOR_PLAIN_DIGESTS])
def test_first_contact_mode_finds_the_convention_of_a_proof_made_with_the_cli_constants(tmp_path, flags):
    """A proof made (by the oracle) with the CLI's constants under one of the alternative readings, written with either setting of the two CBOR
    switches: `--try-conventions` decodes it, names the encodings it met and finds the combination that verifies."""
    rc = va.poseidon_round_constants(source="cli", raw_monty=False, sip_adj0=0xFF)  # NOT the default reading: the search has to find it
    path = _write_proof(tmp_path, rc, flags)
    r = subprocess.run([sys.executable, "-m", "valida_amd.verify_cli", path, "--program", "fib", "--n", "25", "--queries", "4", "--constants", "cli", "--try-conventions"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ACCEPTED under: bare integers canonical | constants cli, raw_monty=False, sip_adj0=0xff | final polynomial not observed" in r.stdout
    assert ("bare integers" in r.stdout.splitlines()[0]) == bool(flags & va.CBOR_CANONICAL_FIELDS)
    assert ("plain [Val; 8]" in r.stdout.splitlines()[0]) == bool(flags & va.CBOR_PLAIN_DIGESTS)
    # the plain mode with the right constants selected by hand is not needed for acceptance above, but with the WRONG ones it must refuse
    r2 = subprocess.run([sys.executable, "-m", "valida_amd.verify_cli", path, "--program", "fib", "--n", "25", "--queries", "4", "--constants", "cli"], capture_output=True, text=True, cwd=ROOT)
    assert r2.returncode == 1 and "REJECTED" in r2.stdout


def test_first_contact_mode_lists_every_rejection_when_nothing_fits(tmp_path):
    rc = va.poseidon_round_constants(seed=12345)  # constants no convention will guess
    path = _write_proof(tmp_path, rc, 0)
    r = subprocess.run([sys.executable, "-m", "valida_amd.verify_cli", path, "--program", "fib", "--n", "25", "--queries", "4", "--constants", "cli", "--try-conventions"],
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 1 and "REJECTED under every combination" in r.stdout and r.stdout.count("->") == 12


def test_bare_integers_read_as_montgomery_words(tmp_path):
    """The one reading the image cannot decide by itself: a writer that emits the raw Montgomery `value` without the struct wrapper."""
    w = va.Workload.fib(25)
    mt, prep = w.main_traces(), w.preprocessed()
    rc = va.poseidon_round_constants()
    words = po.prove_basic(mt, prep[0][1], prep[1][1], rc, num_queries=3).words
    blob = va.proof_cbor(words, va.CBOR_CANONICAL_FIELDS)
    plain, seen = va.proof_from_cbor_ex(blob, False)
    assert np.array_equal(plain, words) and seen == 2 | 4
    monty, _ = va.proof_from_cbor_ex(blob, True)
    assert not np.array_equal(monty, words) and monty.size == words.size
    # reading the canonical image as Montgomery multiplies every field element by 2^-32; lengths and log_degrees stay
    assert monty[0] == words[0] and monty[1] == words[1] and int(monty[2]) == int(words[2]) * pow(1 << 32, va.P - 2, va.P) % va.P
