"""Seeded input generators shared by the parity tests (oracle vs HIP path)."""
import numpy as np

BASES = b"ACGT"


def rand_dna(rng, n):
    return bytes(rng.choice(list(BASES), size=n).tolist()) if n else b""


def mutate(rng, seq, sub=0.01, ins=0.005, dele=0.005):
    out = bytearray()
    for b in seq:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append(int(rng.choice([c for c in BASES if c != b])))
        else:
            out.append(b)
        if rng.random() < ins:
            out.append(int(rng.choice(list(BASES))))
    return bytes(out)


def rand_motif(rng, lo=1, hi=8, allow_n=True):
    n = int(rng.integers(lo, hi + 1))
    m = bytearray(rand_dna(rng, n))
    if allow_n and rng.random() < 0.15:
        m[int(rng.integers(0, n))] = ord("N")
    return bytes(m)


def repeat_allele(rng, motifs, total_len, err=0.02):
    """Concatenated runs of the motifs (N filled uniformly), then mutated."""
    out = bytearray()
    while len(out) < total_len:
        m = motifs[int(rng.integers(0, len(motifs)))]
        copies = int(rng.integers(1, 12))
        for _ in range(copies):
            out += bytes(b if b != ord("N") else int(rng.choice(list(BASES))) for b in m)
        if rng.random() < 0.2:
            out += rand_dna(rng, int(rng.integers(1, 9)))  # interruption -> skip states
    return mutate(rng, bytes(out[:total_len]), err, err / 2, err / 2)
