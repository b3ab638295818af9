"""numpy model of the DEVICE algorithm (test infrastructure only, like everything in oracle/).

It restates, step for step, what suffix_array_amd/csrc does on the GPU -- packed-symbol
initial keys, a stable LSD sort, group-head ranks, and prefix-doubling refinement of the
unresolved groups with the end-of-text rule -- so that the algorithm's logic (not the HIP
code) can be checked on the CPU against oracle_naive_sa / oracle_sais.

The output contract is the reference's: sa[0] = n, sa[1..] sorted suffix offsets
(reference src/saca.rs:9-15), order as in reference src/sa.rs:72-84.
"""
import numpy as np


def alphabet(text: np.ndarray):
    """code table (256 entries), bits per symbol (0 = base-sigma packing), symbols per key, radix"""
    used = np.zeros(256, dtype=bool)
    used[text] = True
    sigma = int(used.sum())
    code = np.cumsum(used) - 1
    se = max(sigma, 2)
    if se & (se - 1) == 0:
        bits = se.bit_length() - 1
        return code.astype(np.uint64), bits, 64 // bits, se
    k = 0
    while se ** (k + 1) <= 2 ** 64:
        k += 1
    return code.astype(np.uint64), 0, k, se


def pack_keys(text: np.ndarray):
    """initial key of every suffix: k symbol codes, most significant first, zero codes past the end"""
    n = text.size
    code, bits, k, se = alphabet(text)
    sym = np.concatenate([code[text], np.zeros(k, dtype=np.uint64)])
    key = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        key = ((key << np.uint64(bits)) | sym[j:j + n]) if bits else (key * np.uint64(se) + sym[j:j + n])
    return key, bits, k


def build(text_bytes: bytes, max_rounds: int = 64, stats: dict | None = None) -> np.ndarray:
    t = np.frombuffer(text_bytes, dtype=np.uint8)
    n = t.size
    out = np.zeros(n + 1, dtype=np.uint32)
    out[0] = n
    if n == 0:
        return out
    key, bits, k = pack_keys(t)
    sa = np.argsort(key, kind="stable").astype(np.int64)
    skey = key[sa]
    head = np.ones(n, dtype=bool)
    head[1:] = skey[1:] != skey[:-1]
    pos = np.arange(n, dtype=np.int64)
    gh = np.maximum.accumulate(np.where(head, pos, 0))          # group head slot of each slot
    isa = np.zeros(n, dtype=np.int64)
    isa[sa] = gh + 1                                             # ranks start at 1
    nxt = np.ones(n, dtype=bool)
    nxt[:-1] = head[1:]
    unresolved = ~(head & nxt)
    U = pos[unresolved]                                          # slots still in groups > 1
    G = gh[unresolved]
    V = sa[U]
    h = k
    rounds = 0
    while U.size:
        rounds += 1
        assert rounds <= max_rounds
        p = V + h
        inside = p < n
        key2 = np.where(inside, n + isa[np.minimum(p, n - 1)], n - 1 - V)   # end-of-text rule
        order = np.lexsort((key2, G))                            # sort by (group head, key2)
        G, key2, V = G[order], key2[order], V[order]
        sa[U] = V
        m = U.size
        nh = np.ones(m, dtype=bool)
        nh[1:] = (G[1:] != G[:-1]) | (key2[1:] != key2[:-1])
        ngh = np.maximum.accumulate(np.where(nh, U, 0))
        isa[V] = ngh + 1
        nn = np.ones(m, dtype=bool)
        nn[:-1] = nh[1:]
        keep = ~(nh & nn)
        U, G, V = U[keep], ngh[keep], V[keep]
        h *= 2
    if stats is not None:
        stats.update(rounds=rounds, bits=bits, k=k)
    out[1:] = sa.astype(np.uint32)
    return out
