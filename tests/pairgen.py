"""Random sequence pairs for the pairwise-alignment tests."""
import numpy as np


def random_seq(rng, n, alphabet=b"ACGT") -> bytes:
    return bytes(rng.choice(list(alphabet), n).tolist())


def mutate(rng, s: bytes, rate: float, alphabet=b"ACGT", long_indels: int = 0) -> bytes:
    """Substitutions, insertions and deletions at `rate` in total, plus `long_indels` block events of 20..300 bases."""
    out = bytearray()
    for ch in s:
        r = rng.random()
        if r < rate / 3:
            continue
        out.append(int(rng.choice(list(alphabet))) if r < 2 * rate / 3 else ch)
        if rng.random() < rate / 3:
            out.append(int(rng.choice(list(alphabet))))
    for _ in range(long_indels):
        a = int(rng.integers(0, max(1, len(out) - 1)))
        k = int(rng.integers(20, 300))
        if rng.random() < 0.5:
            del out[a:a + k]
        else:
            out[a:a] = bytes(rng.choice(list(alphabet), k).tolist())
    if not out:
        out = bytearray(b"A")
    return bytes(out)
