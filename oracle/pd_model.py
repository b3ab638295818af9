"""numpy model of the DEVICE algorithm (test infrastructure only, like everything in oracle/).

It restates the LOGIC of what suffix_array_amd/csrc does on the GPU -- packed-symbol initial keys, a
stable sort, group-head ranks, text-keyed refinement rounds while many suffixes are tied, then
prefix-doubling refinement of the unresolved groups with the end-of-text rule -- so that the
algorithm (not the HIP code) can be checked on the CPU against oracle_naive_sa / oracle_sais.
Device-side staging that does not change the result is not modelled: the 32-bit first stage of the
initial sort and its one-pass finish, the in-LDS group sort versus the global sort, the bit-field
packing of the text-round keys, the sparse rank look-up versus a full ISA, the choice of a group's rank
(first slot here; the device's dense rounds use the last slot: any slot of the group orders the same).

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


def _text_symbols(n: int, bits: int, se: int) -> int:
    """symbols per text-keyed round: what fits below the group head in a 64-bit sort key"""
    room = 64 - max(1, (max(n - 1, 1)).bit_length())
    if bits:
        return room // bits
    s = 0
    while se ** (s + 1) <= 2 ** room:
        s += 1
    return s


def build(text_bytes: bytes, max_rounds: int = 64, stats: dict | None = None, sparse_div: int = 64,
          max_text_rounds: int = 4) -> np.ndarray:
    t = np.frombuffer(text_bytes, dtype=np.uint8)
    n = t.size
    out = np.zeros(n + 1, dtype=np.uint32)
    out[0] = n
    if n == 0:
        return out
    code, bits, k, se = alphabet(t)
    key, bits, k = pack_keys(t)
    sa = np.argsort(key, kind="stable").astype(np.int64)
    skey = key[sa]
    head = np.ones(n, dtype=bool)
    head[1:] = skey[1:] != skey[:-1]
    pos = np.arange(n, dtype=np.int64)
    gh = np.maximum.accumulate(np.where(head, pos, 0))          # group head slot of each slot
    nxt = np.ones(n, dtype=bool)
    nxt[:-1] = head[1:]
    unresolved = ~(head & nxt)
    U = pos[unresolved]                                          # slots still in groups > 1
    G = gh[unresolved]
    V = sa[U]
    depth = k
    rounds = text_rounds = 0

    def refine(key2):
        """sort the tied suffixes by (group head, key2), write them back, re-rank, keep what is still tied"""
        nonlocal U, G, V
        order = np.lexsort((key2, G))
        G2, k2, V2 = G[order], key2[order], V[order]
        sa[U] = V2
        m = U.size
        nh = np.ones(m, dtype=bool)
        nh[1:] = (G2[1:] != G2[:-1]) | (k2[1:] != k2[:-1])
        ngh = np.maximum.accumulate(np.where(nh, U, 0))
        nn = np.ones(m, dtype=bool)
        nn[:-1] = nh[1:]
        keep = ~(nh & nn)
        changed = (V2, ngh)
        U, G, V = U[keep], ngh[keep], V2[keep]
        return changed

    # ---- text-keyed rounds: while many suffixes are tied, the secondary key is the next s symbols of
    # the text (zero codes past the end), so no rank array is needed; depth grows by s per round
    s_sym = _text_symbols(n, bits, se)
    symz = np.concatenate([code[t], np.zeros(s_sym + 1, dtype=np.uint64)])
    progressing = True
    while U.size > n // sparse_div and text_rounds < max_text_rounds and progressing and s_sym > 0:
        before = U.size
        p = V + depth
        tk = np.zeros(U.size, dtype=np.uint64)
        for i in range(s_sym):
            c = symz[np.minimum(p + i, n)]
            tk = ((tk << np.uint64(bits)) | c) if bits else (tk * np.uint64(se) + c)
        refine(tk)
        depth += s_sym
        text_rounds += 1
        rounds += 1
        progressing = U.size * 4 <= before * 3
    # ---- prefix doubling on what is left; ranks of the current order (the device looks them up in the
    # sorted keys / the text, or scatters a full ISA -- same values)
    isa = np.zeros(n, dtype=np.int64)
    isa[sa] = pos + 1                                            # resolved suffix: its slot + 1
    isa[V] = G + 1                                               # tied suffix: slot of its group head + 1
    h = depth
    while U.size:
        rounds += 1
        assert rounds <= max_rounds
        p = V + h
        inside = p < n
        key2 = np.where(inside, n + isa[np.minimum(p, n - 1)], n - 1 - V)   # end-of-text rule
        V2, ngh = refine(key2.astype(np.int64))
        isa[V2] = ngh + 1
        h *= 2
    if stats is not None:
        stats.update(rounds=rounds, text_rounds=text_rounds, bits=bits, k=k)
    out[1:] = sa.astype(np.uint32)
    return out
