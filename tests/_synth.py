"""Small deterministic input generators shared by the tests."""
import json
import os
import random

COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(s: bytes) -> bytes:
    return s.translate(COMP)[::-1]


def rand_dna(rng: random.Random, n: int) -> bytes:
    return bytes(rng.choice(b"ACGT") for _ in range(n))


def fuzz_stream(rng: random.Random, strain: bytes, nreads: int, junk=b"NnRYKMUu-. acgt\rX*", p_junk=0.02,
                min_len=0, max_len=200) -> bytes:
    """Reads drawn from `strain` (either strand), random reads, random junk bytes, any case."""
    out = []
    for _ in range(nreads):
        ln = rng.randint(min_len, max_len)
        if rng.random() < 0.6 and len(strain) > ln + 1:
            a = rng.randrange(0, len(strain) - ln)
            r = bytearray(strain[a:a + ln])
            if rng.random() < 0.5:
                r = bytearray(revcomp(bytes(r)))
        else:
            r = bytearray(rand_dna(rng, ln))
        for i in range(len(r)):
            x = rng.random()
            if x < p_junk:
                r[i] = rng.choice(junk)
            elif x < p_junk * 2:
                r[i] = rng.choice(b"ACGT")
            elif x < p_junk * 3:
                r[i] = r[i] | 0x20
        out.append(bytes(r))
    return b"\n".join(out) + b"\n"



def two_expansions_strain(golden, tmp_path):
    """the 9.2 Mbp strain of tests/golden/make_two_expansions_facts.py, written again by that script's own function; None if this
    numpy draws another sequence than the one the reference saw (the facts carry the file's md5)"""
    import hashlib
    import importlib.util
    facts = json.load(open(os.path.join(golden, "two_expansions_facts.json")))
    spec = importlib.util.spec_from_file_location("make_two_expansions_facts", os.path.join(golden, "make_two_expansions_facts.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    path = str(tmp_path / "strain.fa")
    mod.write_strain(path)
    if hashlib.md5(open(path, "rb").read()).hexdigest() != facts["strain"]["md5"]:
        return None, facts
    return path, facts
